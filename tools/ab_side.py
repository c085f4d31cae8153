"""A/B of the side-stream options: env OFASR_MBCONV_SIDE_STREAM (library) and ops.SIDE_STREAM (static conv backward)."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mb in ("1", "0"):
    for cv in ("1", "0"):
        env = dict(os.environ, OFASR_MBCONV_SIDE_STREAM=mb, OFASR_CONV_SIDE_STREAM=cv)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "8", "--no-cpu-baseline",
                              "--no-roofline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        import json
        j = json.loads(out)
        print("mbconv side=%s conv side=%s  %.1f img/s  %.3f ms/step" % (mb, cv, j["value"], j["ms_per_step"]), flush=True)
