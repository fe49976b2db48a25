"""Two layers in one launch (Engine::PendingConv): logits of ResNet-18 at single-board sizes with the pairing on (default) and off
(CV_PAIR=0, child process), compared bit for bit (developer tool; timing: tools/latency.py with and without CV_PAIR=0)."""
import hashlib, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "chessvision-3lc_amd"))
import torch
from chessvision.hip_backend import HipEngine
from oracle import synth

def run():
    out = {}
    for prec in ("f16x3", "f32", "f16", "f16r"):
        eng = HipEngine(precision=prec)
        eng.load_resnet18(synth.make_resnet(seed=2).state_dict())
        for n in (64, 128, 1, 640):
            x = synth.squares_input(seed=4, n=n).cuda()
            y = eng.resnet18_forward(x)
            out[(prec, n)] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
    return out

if __name__ == "__main__":
    if os.environ.get("PAIR_CHILD"):
        print(repr(run())); sys.exit(0)
    a = run()
    env = dict(os.environ, CV_PAIR="0", PAIR_CHILD="1")
    b = eval(subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    bad = [k for k in a if a[k] != b[k]]
    print("paired vs CV_PAIR=0: differing outputs:", bad, "of", len(a), "(timing: tools/latency.py with and without CV_PAIR=0)")
