# round-1: deferred class-id store in the dual-tile kernel — parity suite, then A/B against variant 2 on the same box
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g inf/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["verified_vs_oracle"])'
for v in 3 2 3 2 3; do timeout 120 python bench.py --no-cpu --variant $v --steps 20 2>/dev/null | tail -1 | python -c "$P" variant$v; done
