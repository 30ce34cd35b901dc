R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
S=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_shape_bench.json 2> $O/driver_shape_bench.err; echo "driver-shape wall $(( $(date +%s) - S )) s" > $O/wall.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
{ echo "# r06, on the round's final build (wavefront-ordered intra launch, groups of up to 128, the next parse behind the last part): tools/exp_refusals.py 1500 --gpu; tools/soak_parity.py 12288 A gop / 8192 B gop / 3000 C gop / 12288 B lockstep; tools/fuzz_intra_gpu.py 4000 100; tools/fuzz_inter_gpu.py 4000 100";
  timeout 900 python tools/exp_refusals.py 1500 --gpu 2>&1 | grep -E "^GPU";
  timeout 600 python tools/soak_parity.py 12288 A gop 2>&1 | tail -2; timeout 600 python tools/soak_parity.py 8192 B gop 2>&1 | tail -2; timeout 600 python tools/soak_parity.py 3000 C gop 2>&1 | tail -2; timeout 600 python tools/soak_parity.py 12288 B lockstep 2>&1 | tail -2;
  timeout 600 python tools/fuzz_intra_gpu.py 4000 100 2>&1 | tail -2; timeout 600 python tools/fuzz_inter_gpu.py 4000 100 2>&1 | tail -2; } > $O/long_fuzz.txt 2>&1
