# round-1 experiment: L2 prefetch-touch of the tile after next (variant 4) vs the default dual-tile kernel (variant 3)
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g inf/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["verified_vs_oracle"])'
for v in 3 4 3 4; do timeout 40 python bench.py --no-cpu --variant $v --steps 10 2>/dev/null | tail -1 | python -c "$P" variant$v; done
