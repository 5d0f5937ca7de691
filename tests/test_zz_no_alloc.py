"""
GPU: hk_mog_apply_batch_static -- the batch launch whose scratch (learning-rate
table, launch-order lists) is a static device array of the library, for
callers that cannot allocate -- against hk_mog_apply_batch, bit for bit (masks
and model), on two calls in a row, the second with a smaller input.

Collected last (file name): the round's end runs pytest -x.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

AOS_FIELD = (1, 0, 2, 3, 4, 5, 6, 7)    # MixData order -> plane field


def frames_moving(rng, t, h, w):
    base = rng.randint(30, 200, size=(h, w, 3)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.empty((t, h, w, 3), np.uint8)
    for i in range(t):
        f = base + rng.normal(0, 3.0, size=base.shape)
        for d in range(3):
            cx = (7 + 11 * d + 2.5 * i) % w
            cy = (5 + 9 * d + 1.5 * i * (d + 1)) % h
            f[(xx - cx)**2 + (yy - cy)**2 <= 9 + 4 * d] = 240 - 60 * d
        out[i] = np.clip(f, 0, 255).astype(np.uint8)
    return out


def planes_to_aos(planes, n):
    p = planes.cpu().numpy().reshape(8, 5, n)           # [field][k][pixel]
    return np.ascontiguousarray(
        p[list(AOS_FIELD)].transpose(2, 1, 0))          # [pixel][k][field]


def static_batches(frames, splits, w, h):
    """The frames through the static entry point in the given batches, on a
    caller-owned model; returns masks and the model in AoS order."""
    from hawkeye_amd import _native as nat
    n = w * h
    model = torch.zeros((40, n), dtype=torch.float32, device='cuda')
    masks = torch.empty(frames.shape[:3], dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    for s, e in splits:
        part, out = frames[s:e], masks[s:e]
        nat.check(nat.lib.hk_mog_apply_batch_static(
            w, h, 0, model.data_ptr(), s, part.data_ptr(), part.stride(0),
            part.stride(1), out.data_ptr(), out.stride(0), out.stride(1),
            e - s, stream))
    torch.cuda.synchronize()
    return masks.cpu().numpy(), planes_to_aos(model, n)


def handle_batches(frames, splits, w, h):
    from hawkeye_amd.detect import BackgroundSubtractorMOG
    mog = BackgroundSubtractorMOG(w, h)
    masks = np.concatenate([mog.apply_batch(frames[s:e]).cpu().numpy()
                            for s, e in splits])
    model = mog.export_model()
    mog.close()
    return masks, model


def oracle_batches(host_frames, w, h):
    from oracle import detect_ref as D
    ref = D.Mog(w, h)
    masks = np.stack([ref.apply(f) for f in host_frames])
    return masks, ref.model()


def compare(seed, t, h, w, splits):
    rng = np.random.RandomState(seed)
    host = frames_moving(rng, t, h, w)
    frames = torch.from_numpy(host).cuda()
    want_masks, want_model = handle_batches(frames, splits, w, h)
    got_masks, got_model = static_batches(frames, splits, w, h)
    assert 0 < (want_masks == 255).mean() < 0.9
    assert np.array_equal(got_masks, want_masks)
    assert np.array_equal(got_model.view(np.uint32),
                          want_model.view(np.uint32))
    # and against the C oracle itself, so that this file stands alone
    ref_masks, ref_model = oracle_batches(host, w, h)
    assert np.array_equal(got_masks, ref_masks)
    assert np.array_equal(got_model.view(np.uint32),
                          ref_model.view(np.uint32))


def test_static_scratch_entry_point_equals_the_allocating_one():
    # call 1: 128x96 (tiled kernel), two batches on one model -- the second
    # is ordered by the costs the first left in the static array
    compare(11, 40, 96, 128, [(0, 24), (24, 40)])
    # call 2, right after it, smaller input, fresh model: the array still
    # holds the costs and the order of the larger geometry
    compare(12, 18, 48, 64, [(0, 18)])
    # and an odd geometry (general kernel), smaller again
    compare(13, 9, 37, 53, [(0, 9)])
    # continuing batches on the odd geometry, then back to the tiled one on
    # the same stream: the costs in the array belong to another kernel's units
    compare(14, 20, 37, 53, [(0, 7), (7, 20)])
    compare(15, 12, 96, 128, [(0, 5), (5, 12)])


def test_static_scratch_entry_point_takes_no_device_memory():
    """Its defining property: after the first call on the device (which looks
    the array's address up) no call changes the amount of free device memory,
    whatever the geometry -- a hipMalloc inside would, cached or not."""
    from hawkeye_amd import _native as nat
    rng = np.random.RandomState(21)
    shapes = [(24, 48, 64), (16, 96, 128), (9, 37, 53), (6, 270, 480)]
    inputs = []
    for t, h, w in shapes:          # every torch allocation up front
        frames = torch.from_numpy(frames_moving(rng, t, h, w)).cuda()
        inputs.append((frames, torch.zeros((40, w * h), dtype=torch.float32,
                                           device='cuda'),
                       torch.empty(frames.shape[:3], dtype=torch.uint8,
                                   device='cuda')))
    stream = torch.cuda.current_stream().cuda_stream

    def run(k, first):
        frames, model, masks = inputs[k]
        t, h, w = shapes[k]
        nat.check(nat.lib.hk_mog_apply_batch_static(
            w, h, 0, model.data_ptr(), first, frames.data_ptr(),
            frames.stride(0), frames.stride(1), masks.data_ptr(),
            masks.stride(0), masks.stride(1), t, stream))

    run(0, 0)                       # first call on the device
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for rep in range(3):
        for k in range(len(shapes)):
            run(k, rep * shapes[k][0])
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before


def test_static_scratch_entry_point_rejects_frames_beyond_its_maximum():
    from hawkeye_amd import _native as nat
    rc = nat.lib.hk_mog_apply_batch_static(4096, 4096, 0, 8, 0, 8, 0, 3 * 4096,
                                           8, 0, 4096, 1, None)
    assert rc == nat.HK_ERR_CAPACITY
    assert b'exceed' in nat.lib.hk_last_error()
