REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj
cp $P/libmobiclip_hip.so /tmp/lib_keep.so
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o /tmp/rgbv.o $O/mobi_dparse.hip.o $O/mobi_lsparse.hip.o $O/mobi_analysis.hip.o"
for ROUND in 1 2; do
for V in asis nt; do
  cp $P/csrc/mobi_rgb.hip /tmp/mobi_rgb_v.hip
  if [ $V = nt ]; then
    sed -i 's|    \*(uint4 \*)o = uint4{pe\[0\], pe\[1\], pe\[2\], pe\[3\]};|    __builtin_nontemporal_store(u4{pe[0], pe[1], pe[2], pe[3]}, (u4 *)o);|; s|    \*(uint4 \*)(o + width) = uint4{po\[0\], po\[1\], po\[2\], po\[3\]};|    __builtin_nontemporal_store(u4{po[0], po[1], po[2], po[3]}, (u4 *)(o + width));|; s|typedef float f32x2 __attribute__((ext_vector_type(2)));|typedef float f32x2 __attribute__((ext_vector_type(2)));\ntypedef uint32_t u4 __attribute__((ext_vector_type(4)));|' /tmp/mobi_rgb_v.hip
  fi
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -fvisibility=hidden -I$P/csrc -c /tmp/mobi_rgb_v.hip -o /tmp/rgbv.o 2>&1 | grep -E " error" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$O/exports.map $OBJS -o $P/libmobiclip_hip.so || exit 1
  echo "[$V] $(timeout 120 python $REPO/tools/exp_rgb.py 2>&1 | tail -1)"
done
done
cp /tmp/lib_keep.so $P/libmobiclip_hip.so
