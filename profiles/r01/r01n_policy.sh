# round-1 experiment: cache-policy bits on the image LDS-DMA loads of the dual-tile kernel
# variant 3 = nt (default), 4 = sc1 nt, 5 = sc0 sc1 nt, 6 = sc1, 7 = none
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g inf/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["verified_vs_oracle"])'
for rep in 1 2; do for v in 3 4 5 6 7; do timeout 120 python bench.py --no-cpu --variant $v --steps 20 2>/dev/null | tail -1 | python -c "$P" variant$v; done; done
