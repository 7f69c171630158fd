"""Local search over AVK_STREAM_ORDER (the order the context's side streams are made in; x = a stream nothing is queued on): the runtime hands streams their hardware
queues in creation order, and which launches share a queue's pipe is worth 10-40 % of a step.  Every candidate runs in fresh processes (tools/gpu_enqueue_time.py: the
shard of an 8-rank job and the whole genome, 100 queued resident steps, best of 3).  usage on the GPU box: python tools/stream_order_search.py [seconds=400] [start order]"""
import os, random, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
start = sys.argv[2] if len(sys.argv) > 2 else "stwxxabxxcd"
base = {"8": 1.09, "1": 2.63, "b1": 6.9, "b8": 1.8}
cache = {}


def step_ms(order, ranks):
    env = dict(os.environ, AVK_STREAM_ORDER=order)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_enqueue_time.py"), ranks], env=env, capture_output=True, text=True, timeout=120).stdout
    v = [float(m) for m in re.findall(r"\(([0-9.]+) ms per step\)\s*$", out, re.M)]
    return min(v) if v else 99.0


def call_ms(order, ranks):
    env = dict(os.environ, AVK_STREAM_ORDER=order)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_boundary_time.py"), ranks, "30", "kernel_copies=2"], env=env, capture_output=True, text=True, timeout=120).stdout
    v = [float(m) for m in re.findall(r"median ([0-9.]+)", out)]
    return min(v) if v else 99.0


def score(order):
    if order not in cache:
        s, g, b1, b8 = step_ms(order, "8"), step_ms(order, "1"), call_ms(order, "1"), call_ms(order, "8")
        cache[order] = ((s / base["8"]) * (g / base["1"]) * (b1 / base["b1"]) ** 2 * (b8 / base["b8"])) ** 0.2, s, g, b1, b8
        print("%-16s resident shard %.3f genome %.3f | call (copies by kernel) genome %.3f shard %.3f | score %.4f" % (order, s, g, b1, b8, cache[order][0]), flush=True)
    return cache[order][0]


def neighbours(o):
    out = set()
    for i in range(len(o) + 1):
        if o.count("x") < 6:
            out.add(o[:i] + "x" + o[i:])
    for i, ch in enumerate(o):
        if ch == "x":
            out.add(o[:i] + o[i + 1:])
    for i in range(len(o) - 1):
        if o[i] != o[i + 1]:
            out.add(o[:i] + o[i + 1] + o[i] + o[i + 2:])
    out.discard(o)
    return sorted(out)


rng = random.Random(7)
t0 = time.time()
best = start
score(best)
while time.time() - t0 < budget:
    cand = [c for c in neighbours(best) if c not in cache]
    if not cand:
        break
    c = rng.choice(cand)
    if score(c) < cache[best][0] - 0.003:
        best = c
        print("  -> best so far %s" % best, flush=True)
print("best %s: resident shard %.3f genome %.3f, call genome %.3f shard %.3f" % ((best,) + cache[best][1:]))
