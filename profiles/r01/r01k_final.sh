# round-1 final verification on the GPU box: full GPU suite, smoke, bench rows for DESIGN.md §8, QAT op probe
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g inf/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel"], d["verified_vs_oracle"])'
timeout 300 python bench.py > gpurun_out/bench_r01k.json 2>/dev/null; python -c "$P" default < gpurun_out/bench_r01k.json
timeout 120 python bench.py --no-cpu --dist M 2>/dev/null | tail -1 | python -c "$P" distM
timeout 120 python bench.py --no-cpu --logits 2>/dev/null | tail -1 | python -c "$P" logits
timeout 120 python bench.py --no-cpu --model mcu_1k 2>/dev/null | tail -1 | python -c "$P" mcu_1k
timeout 120 python bench.py --no-cpu --model mcu_12k_fp130 2>/dev/null | tail -1 | python -c "$P" fp130
timeout 120 python bench.py --no-cpu --model tern_96 --path 1 2>/dev/null | tail -1 | python -c "$P" tern_mfma
timeout 120 python profiles/qat_bench.py 2>&1 | grep -v amdgpu.ids
