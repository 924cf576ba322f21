"""Random search over the cost table that deals k_amaze_rows' items to its waves (MLVFS_AMD_AMAZE_ROWS_COSTS): each candidate is timed
with tools/amaze_rows_time.py in a fresh process (the table is read once per process).  usage: amaze_rows_cost_search.py [candidates]"""
import os, random, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
random.seed(int(os.environ.get("SEED", "1")))
def run(costs):
    env = dict(os.environ, MLVFS_AMD_AMAZE_ROWS_COSTS=",".join(str(c) for c in costs), MLVFS_AMD_AMAZE_ROWS_SKIP=os.environ.get("MLVFS_AMD_AMAZE_ROWS_SKIP", "4"))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "amaze_rows_time.py")], env=env, capture_output=True, text=True).stdout
    m = re.search(r"([0-9.]+) ms \(median", out)
    return float(m.group(1)) if m else 1e9
best = [1000] * 18
best_t = run(best)
print("uniform", best_t, flush=True)
for k in range(N):
    cand = [max(100, int(c * random.choice([0.6, 0.8, 1.0, 1.0, 1.25, 1.6]))) for c in best] if k % 2 else [random.choice([600, 1000, 1500, 2200, 3000]) for _ in best]
    t = run(cand)
    if t < best_t:
        best, best_t = cand, t
        print(k, t, cand, flush=True)
print("best", best_t, best)
