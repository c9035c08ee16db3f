import subprocess, sys, os, json
here = os.path.dirname(os.path.abspath(__file__))
replay = os.path.join(os.path.dirname(here), "poly_replay.py")
path = sys.argv[1]
def run(n, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, replay, "child", path, str(n)], env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads(r.stdout.strip().splitlines()[-1])
lo, hi = int(sys.argv[2]), int(sys.argv[3])       # equal at lo, different at hi
while hi - lo > 1:
    mid = (lo + hi) // 2
    a, b = run(mid, {}), run(mid, {"BSLV_K2_LDS": "64"})
    same = a["digest"] == b["digest"]
    print(mid, same, a["slots"], b["slots"], a["edges"], b["edges"], flush=True)
    if same: lo = mid
    else: hi = mid
print("first differing prefix length:", hi)
