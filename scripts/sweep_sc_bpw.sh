for bpw in 1 2 4 8; do
  echo "== BPW=$bpw"
  NALO_SC_BPW=$bpw python bench.py --workload stress250k --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stress', d['value'], d['kernel_ms']['ba_sc'])"
  for P in 250000 1000000; do NALO_SC_BPW=$bpw python scripts/run_shard_leg.py $P 2>&1 | grep -oE "keyframes_per_s.: [0-9.]+|.ba_sc_us.: [0-9.]+" | tr '\n' ' '; echo; done
done
