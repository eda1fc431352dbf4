"""Same-box A/B of several builds of the library on BASELINE configs[1]: python tools/ab_libs.py [reps=2] [f32] [option=value ...] lib1.so lib2.so ...
('prod' = the committed gato_python_amd/libgato_hip.so).  Every build runs in a child process of its own (tools/ab_one.py),
alternating, so box-to-box and drift effects cancel."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so") or a == "prod"]
opts = [a for a in sys.argv[1:] if a not in libs and not a.startswith("reps=")]
reps = ([int(a[5:]) for a in sys.argv[1:] if a.startswith("reps=")] or [2])[0]
acc = {l: [] for l in libs}
for rep in range(reps):
    for l in libs:
        env = dict(os.environ)
        if l != "prod": env["GATO_HIP_LIB"] = os.path.join(root, l) if not os.path.isabs(l) else l
        else: env.pop("GATO_HIP_LIB", None)
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_one.py")] + opts, env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", p.stderr[-800:], flush=True); continue
        r = json.loads(line[-1]); acc[l].append(r)
        print(f"rep {rep} {os.path.basename(l):24s} {json.dumps(r)}", flush=True)
print("--- medians")
import statistics as st
for l in libs:
    if acc[l]:
        print(f"{os.path.basename(l):24s} us/iter {st.median(r['us_per_iter'] for r in acc[l]):.4f}  us/step {st.median(r['us_per_step'] for r in acc[l]):.2f}"
              f"  rel_lam_12 {max(r['rel_lam_12'] for r in acc[l]):.2e} iters {acc[l][0]['iters']}/{acc[l][0]['iters_oracle']}")
