"""vx_yolo_pipeline_step through YoloV3PipelineStep: same bits as the two module calls, over more calls than the ring
has slots, for two batch sizes and a resized input."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,hw,sthr", [(1, (416, 416), 0.9), (3, (416, 416), 0.3), (2, (360, 640), 0.5)])
def test_one_call_step_equals_the_two_module_calls(B, hw, sthr):
    import bench
    import vortex_amd as V
    from vortex_amd import ops
    dev = torch.device("cuda", 0)
    S = 416
    norm = V.Normalize().to(dev)
    post = V.YoloV3FusedPostProcess(img_size=S)
    step = V.YoloV3PipelineStep(norm, post, depth=2)
    st, it = torch.tensor([sthr]), torch.tensor([0.45])
    for k in range(5):
        _, heads = bench.synth_on_device(B, dev, seed=700 + k)
        g = torch.Generator(device=dev).manual_seed(900 + k)
        img = torch.randint(0, 256, (B, hw[0], hw[1], 3), generator=g, dtype=torch.uint8, device=dev)
        x, res = step(img, heads, st, it)
        mean, std, scaler = norm._consts()
        want_x = norm(img) if hw == (S, S) else ops.preprocess(img, mean, std, scaler, out_size=(S, S))
        want = post(heads, st, it)
        assert torch.equal(x, want_x)
        assert len(res) == len(want) == B
        for a, b in zip(res, want):
            assert torch.equal(a, b)
    assert step(torch.zeros((0, S, S, 3), dtype=torch.uint8, device=dev), heads, st, it)[1] == ()
