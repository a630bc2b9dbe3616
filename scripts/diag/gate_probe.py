"""the gate-cancel flow of tests/test_ba_gpu.py as a probe: N child processes per mode (plain / cancelled first optimize), the poses after restore + optimize hashed;
every line should be the same. usage: gate_probe.py [repeats]   (through scripts/ab.sh for library variants)"""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
src = open(os.path.join(ROOT, "tests", "test_ba_gpu.py")).read()
script = src[src.index('_GATE_CANCEL_SCRIPT = r"""') + len('_GATE_CANCEL_SCRIPT = r"""'):]
script = script[:script.index('"""')]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "gc.py"); open(p, "w").write(script)
    seen = {}
    for rep in range(n):
        for name, extra in (("plain", {}), ("cancel", {"NALO_BA_TEST_GATE_CANCEL": "1"})):
            env = dict({k: v for k, v in os.environ.items() if k != "NALO_BA_TEST_GATE_CANCEL"}, **extra)
            r = subprocess.run([sys.executable, p, ROOT], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(name, "FAILED", r.stderr[-300:]); continue
            d = json.loads(line[-1][7:])
            h = hashlib.md5(json.dumps([d["rmse_after"], d["w2c_after"]]).encode()).hexdigest()[:10]
            seen.setdefault(h, []).append(name)
            print(rep, name, d["first"][:40], "rmse %.9g" % d["rmse_after"], h, flush=True)
    print("distinct results:", {k: len(v) for k, v in seen.items()})
