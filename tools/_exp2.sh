cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_lsparse_gpu.py tests/test_device_parse.py tests/test_parse_fallback.py -m gpu -x -q 2>&1 | tail -5
export MOBI_LIB=$GRAFT_REPO_ROOT/mobiclipdecoder_amd/libmobiclip_hip_prof.so
for N in 1024 2048 4096 6144 8192 12288 24576 49152; do
  echo "== $N clips"; timeout 900 python tools/exp_dparse.py $N --lockstep 2>&1 | grep clips= | grep -o "inside the C call.*staging [0-9.]* ms"
  LOCKSTEP=1 timeout 900 python tools/exp_async.py $N 8 2>&1 | grep -E "^asynchronous|rror"
done
