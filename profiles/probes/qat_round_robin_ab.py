import os, sys, subprocess, json
repo = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for rows in (1000000, 10000000):
    res = {"0": [], "1": []}
    for rep in range(4):
        for flag in ("0", "1"):
            r = subprocess.run([sys.executable, os.path.join(repo, "profiles/qat_model_bench.py"), "--rows", str(rows), "--steps", "30"], capture_output=True, text=True, env=dict(os.environ, BNM_QAT_RR_PROBE=flag))
            res[flag].append(round(json.loads(r.stdout.strip().splitlines()[-1])["hbm_frac"], 4))
    print(rows, "dynamic", res["0"], "round-robin", res["1"])
