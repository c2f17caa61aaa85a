# round-1 experiment: dual-tile variant (3) vs two-in-flight variant (2), class-count-specialised argmax
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"])'
for v in 2 3 2 3; do timeout 120 python bench.py --variant $v --no-cpu --steps 20 2>&1 | tail -1 | python -c "$P" "variant$v"; done
for v in 2 3; do BNM_DIAG_SRC_WRAP=256 timeout 120 python bench.py --variant $v --no-cpu --no-verify --steps 20 2>&1 | tail -1 | python -c "$P" "compute-only-variant$v"; done
timeout 120 python bench.py --model mcu_1k --no-cpu --steps 20 2>&1 | tail -1 | python -c "$P" "mcu_1k-default"
timeout 120 python profiles/stream_ceiling.py 2>&1 | tail -12
