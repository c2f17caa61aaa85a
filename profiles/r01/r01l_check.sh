# round-1: QAT conv op, dual kernel on the 1k model, Dist-M row
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g inf/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["fused_variant"], d["verified_vs_oracle"])'
timeout 120 python bench.py --no-cpu --dist 1 2>/dev/null | tail -1 | python -c "$P" distM
for v in 1 3 1 3; do timeout 120 python bench.py --no-cpu --model mcu_1k --variant $v --steps 20 2>/dev/null | tail -1 | python -c "$P" mcu_1k_v$v; done
timeout 120 python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "$P" default
