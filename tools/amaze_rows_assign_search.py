"""Local search over the assignment of k_amaze_rows' items to its 16 waves (MLVFS_AMD_AMAZE_ROWS_ASSIGN), objective = the kernel's
time on a plane of 2 254 complete tiles (tools/amaze_rows_time.py in a fresh process per candidate).
usage: amaze_rows_assign_search.py [evaluations]"""
import os, random, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
random.seed(int(os.environ.get("SEED", "1")))
LOAD = 0
OBJ = os.environ.get("OBJECTIVE", "tiles")          # tiles: the kernel alone on 2 254 tiles; batch: a batch of 8 dual-ISO conversions
def run(assign=None, show=False):
    env = dict(os.environ)
    if OBJ == "tiles": env["MLVFS_AMD_AMAZE_ROWS_SKIP"] = os.environ.get("MLVFS_AMD_AMAZE_ROWS_SKIP", "4")
    if assign is not None:
        env["MLVFS_AMD_AMAZE_ROWS_ASSIGN"] = ",".join(encode(ph, w) for ph in assign for w in ph)
    if show:
        env["MLVFS_AMD_AMAZE_ROWS_SHOW"] = "1"
    if OBJ == "batch":
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "dualiso_batch_bench.py"), "8", "5"], env=env, capture_output=True, text=True)
        m = re.search(r"best ([0-9.]+)\)", r.stderr)
        return (8e3 / float(m.group(1)) if m else 1e9), r.stderr       # ms per batch, best of 5
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "amaze_rows_time.py")], env=env, capture_output=True, text=True)
    m = re.search(r"([0-9.]+) ms \(median", r.stdout)
    return (float(m.group(1)) if m else 1e9), r.stderr
def encode(ph, items):
    v = 0xFFFFFFFFFFFFFFFF
    for k, it in enumerate(items):
        v = (v & ~(0xFFFF << (16 * k))) | (it << (16 * k))
    return "%016x" % v
def decode(word):
    v = int(word, 16); items = []
    for k in range(4):
        it = (v >> (16 * k)) & 0xFFFF
        if it != 0xFFFF: items.append(it)
    return items
t0, err = run(show=True)
line = [l for l in err.splitlines() if re.fullmatch(r"[0-9a-f,]+", l.strip())][-1]
words = line.strip().split(",")
assign = [[decode(w) for w in words[:16]], [decode(w) for w in words[16:]]]
best, best_t = assign, t0
print("start", t0, flush=True)
for k in range(N):
    cand = [[list(w) for w in ph] for ph in best]
    ph = random.randrange(2)
    a, b = random.sample(range(16), 2)
    kind = random.random()
    if kind < 0.15 and len(cand[ph][a]) > 1:                          # another order of a wave's items
        random.shuffle(cand[ph][a])
    elif kind < 0.55 and cand[ph][a]:                                 # move an item
        if len(cand[ph][b]) < 4: cand[ph][b].append(cand[ph][a].pop(random.randrange(len(cand[ph][a]))))
    elif cand[ph][a] and cand[ph][b]:                                  # swap two items
        i, j = random.randrange(len(cand[ph][a])), random.randrange(len(cand[ph][b]))
        cand[ph][a][i], cand[ph][b][j] = cand[ph][b][j], cand[ph][a][i]
    t, _ = run(cand)
    if k % 20 == 19: print("...", k + 1, "candidates, best", best_t, flush=True)      # (a silent run is taken for a hung one)
    if t < best_t - (0.02 if OBJ == "batch" else 0.005):
        best, best_t = cand, t
        print(k, t, flush=True)
print("best", best_t)
print(",".join(encode(ph, w) for ph in best for w in ph))
