import sys, time, cProfile, pstats, tempfile
sys.path.insert(0,'chessvision-3lc_amd'); sys.path.insert(0,'.')
import numpy as np
from chessvision import ChessVision, synthetic
d=tempfile.mkdtemp(); pe,pc=synthetic.save_checkpoints(d)
cv=ChessVision(board_extractor_weights=str(pe), classifier_weights=str(pc), precision="f16x3")
rng=np.random.default_rng(0)
images=[rng.integers(0,256,(512,512,3),dtype=np.uint8) for _ in range(64)]
cv.process_images(images[:8], fallback_quad=True)
pr=cProfile.Profile(); pr.enable()
res=cv.process_images(images, fallback_quad=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
