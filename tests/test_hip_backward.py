"""GPU parity of the backward building blocks (SURVEY.md section 2b "B*": train/train.py:118 loss.backward()) against
torch autograd / plain torch fp32 on the same inputs.  Everything goes through the C ABI (peekvit_amd.ops)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from peekvit_amd import ops as o
    return o


def _bf(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


@pytest.mark.parametrize("R,C", [(64, 64), (197, 128), (1000, 72), (4096, 768), (33, 7)])
def test_transpose_exact(ops, R, C):
    x = _bf(R, C, seed=R + C)
    y = ops.transpose(x)
    assert torch.equal(y, x.t().contiguous())


@pytest.mark.parametrize("R,C,dt", [(5, 8, torch.float32), (1024, 768, torch.bfloat16), (4099, 128, torch.float32), (20000, 3072, torch.bfloat16)])
def test_colsum(ops, R, C, dt):
    x = _bf(R, C, seed=R).to(dt)
    out = torch.full((C,), 3.0, device="cuda")
    ops.colsum(x, out)
    ref = x.double().sum(0)
    assert rel_l2(out.double(), ref) < 2e-6
    ops.colsum(x, out, accumulate=True)
    assert rel_l2(out.double(), 2 * ref) < 2e-6


def test_sum_slices(ops):
    p = torch.randn(5, 384, 128, device="cuda")
    out = torch.ones(384, 128, device="cuda")
    ops.sum_slices(p, out)
    assert rel_l2(out, p.sum(0)) < 1e-6
    ops.sum_slices(p, out, accumulate=True)
    assert rel_l2(out, 2 * p.sum(0)) < 1e-6


@pytest.mark.parametrize("M,No,Ni,ksplit", [(1024, 128, 128, 4), (4096, 768, 768, 8), (8192, 2304, 768, 0), (2048, 384, 1536, 2),
                                            (788, 128, 512, 0)])
def test_wgrad_split_k(ops, M, No, Ni, ksplit):
    """dW = dY^T . X on transposed bf16 operands, split-K slices + reduction, vs fp32 matmul of the same bf16 values."""
    dy, x = _bf(M, No, seed=1, scale=0.1), _bf(M, Ni, seed=2)
    if M % 64:                                               # K (= M) must be a multiple of 64: zero-pad the transposed operands
        pad = 64 - M % 64
        dy = torch.cat([dy, torch.zeros(pad, No, device="cuda", dtype=torch.bfloat16)])
        x = torch.cat([x, torch.zeros(pad, Ni, device="cuda", dtype=torch.bfloat16)])
    out = torch.zeros(No, Ni, device="cuda")
    ops.wgrad(ops.transpose(dy), ops.transpose(x), out, ksplit=ksplit)
    ref = dy.float().t() @ x.float()
    assert rel_l2(out, ref) < 2e-6
    ops.wgrad(ops.transpose(dy), ops.transpose(x), out, accumulate=True, ksplit=ksplit)
    assert rel_l2(out, 2 * ref) < 2e-6
