cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_lsparse_gpu.py tests/test_device_parse.py tests/test_parse_fallback.py tests/test_internal_walk.py -m gpu -x -q 2>&1 | tail -3
export MOBI_LIB=$GRAFT_REPO_ROOT/mobiclipdecoder_amd/libmobiclip_hip_prof.so
for N in 2048 8192 24576 49152; do
  echo "== $N clips"; timeout 900 python tools/exp_dparse.py $N --lockstep 2>&1 | grep clips= | grep -o "inside the C call.*staging [0-9.]* ms"
  LOCKSTEP=1 timeout 900 python tools/exp_async.py $N 8 2>&1 | grep -E "^asynchronous|rror"
done
timeout 900 python tools/soak_parity.py 8192 B lockstep | tail -1
timeout 900 python tools/exp_refusals.py 800 --gpu
