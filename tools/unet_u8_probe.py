import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo/chessvision-3lc_amd"); sys.path.insert(0, "/root/repo")
from chessvision import synthetic, classical
from chessvision.hip_backend import HipEngine
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for seg in (False, True):
    eng = HipEngine(precision="f16x3")
    eng.load_unet(synthetic.unet_state_dict(1, segmenting=seg))
    photos = np.stack([classical.resize_area(synthetic.board_photo(s), (256, 256)) for s in range(64)])
    noise = np.random.default_rng(0).integers(0, 256, photos.shape, dtype=np.uint8)
    for name, arr in (("photos", photos), ("noise", noise)):
        u8 = torch.from_numpy(arr).cuda()
        f32 = (u8.float() / 255).permute(0, 3, 1, 2).contiguous()
        print(f"segmenting={seg} {name}: u8+mask {timeit(lambda: eng.unet_forward_u8(u8, want_mask=True)):.2f} ms  u8 {timeit(lambda: eng.unet_forward_u8(u8, want_mask=False)):.2f} ms  f32 {timeit(lambda: eng.unet_forward(f32, check=False)):.2f} ms per 64 boards", flush=True)
    eng.close()
