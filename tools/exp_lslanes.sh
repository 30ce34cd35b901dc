REPO=${GRAFT_REPO_ROOT:-$(pwd)}; P=$REPO/mobiclipdecoder_amd; O=$P/_obj_prof
cp $P/libmobiclip_hip_prof.so /tmp/lib_keep.so
OBJS="$O/mobi_abi.cpp.o $O/mobi_parse.cpp.o $O/mobi_demux.cpp.o $O/mobi_moflex.cpp.o $O/mobi_kernels.hip.o $O/mobi_rgb.hip.o $O/mobi_dparse.hip.o /tmp/lsv.o $O/mobi_analysis.hip.o"
for N in 64 16 8 4 2 1; do
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 -fvisibility=hidden -DMOBI_PROFILING -mllvm -amdgpu-sched-strategy=max-ilp -DMOBI_LS_SYNC_CLIPS=$N -c $P/csrc/mobi_lsparse.hip -o /tmp/lsv.o 2>&1 | grep -E " error" | head -3
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$O/exports.map $OBJS -o $P/libmobiclip_hip_prof.so || exit 1
  C=$((N * 512)); [ $C -lt 64 ] && C=64
  echo "== $N clips per wave, $C clips: $(timeout 600 python $REPO/tools/exp_dparse.py $C --lockstep 2>&1 | grep clips= | grep -o 'parse kernel: I [0-9.]* ms, P median [0-9.]* ms')"
done
cp /tmp/lib_keep.so $P/libmobiclip_hip_prof.so
