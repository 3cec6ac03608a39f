"""One line per run: ms per step, tagged-kernel sum and self-attention ms per launch of `python bench.py --no-cpu-baseline --no-vae ARGS`
(for tools/ab_variants.py: the same bench under two builds of the library on one box)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-vae"] + sys.argv[1:], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
k = d.get("kernel_ms_per_step") or {}
print(f"ms_per_step {d['ms_per_step']:.2f}  rmsnorm_rope {k.get('rmsnorm_rope')}  ln_modulate {k.get('ln_modulate')}  tagged kernels {sum((d.get('kernel_ms_per_step') or {}).values()):.2f}  flash ms/launch {(d.get('roofline') or {}).get('ms_per_launch')}")
