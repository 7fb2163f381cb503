for i in 1 2; do
  for v in nt nont; do
    if [ $v = nont ]; then export VLB_LIB_PATH=$PWD/vl-bert_amd/csrc/ab0/libvlbert_hip.so; else unset VLB_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-phase-times 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['loss'])"
  done
done
