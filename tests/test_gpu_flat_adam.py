"""GPU: favae_step.FlatAdam -- the one-line swap for torch.optim.Adam in a training loop that is not TrainStep (the reference's own
train(), favae_scripts/train_favae.py:68-119, 297-305): same trajectory as torch.optim.Adam, its state_dict format, views that survive
zero_grad(), and the direct-accumulation mode against ordinary autograd gradients on the drop-in modules."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (64,), (32, 64, 1, 1), (7,), (128, 16)]
    main = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    main[0].data = main[0].data.contiguous(memory_format=torch.channels_last)           # conv weights live channels-last (OHWI)
    extra = [torch.nn.Parameter(torch.randn(4, generator=g).to(DEV))]
    return main, extra


def _grads(ps, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*p.shape, generator=g).to(DEV) for p in ps]


def test_flat_adam_follows_torch_adam_and_speaks_its_state_dict():
    from favae_step import FlatAdam
    m0, e0 = _params(0)
    m1, e1 = copy.deepcopy(m0), copy.deepcopy(e0)
    m1[0].data = m1[0].data.contiguous(memory_format=torch.channels_last)
    ref = torch.optim.Adam([{"params": m0}, {"params": e0, "lr": 2.0e-3}], lr=1.0e-2, betas=(0.5, 0.9))      # train_favae.py:297-299
    opt = FlatAdam([{"params": m1}, {"params": e1, "lr": 2.0e-3}], lr=1.0e-2, betas=(0.5, 0.9))
    assert [g["lr"] for g in opt.param_groups] == [1.0e-2, 2.0e-3]
    for it in range(4):
        ref.zero_grad()
        opt.zero_grad(set_to_none=True)
        for p, q, g in zip(m0 + e0, m1 + e1, _grads(m0 + e0, 10 + it)):
            p.grad = g.clone()
            assert q.grad is not None and float(q.grad.abs().max()) == 0.0, "zero_grad keeps the (zeroed) views"
            q.grad.add_(g)                                       # what AccumulateGrad does with an existing .grad
        ref.step()
        opt.step()
        for p, q in zip(m0 + e0, m1 + e1):
            assert (p - q).abs().max() <= 2e-6 * p.abs().max(), it
    # torch.optim.Adam's format, both directions
    sd = opt.state_dict()
    ref2 = torch.optim.Adam([{"params": copy.deepcopy(m0)}, {"params": copy.deepcopy(e0), "lr": 2.0e-3}], lr=1.0e-2, betas=(0.5, 0.9))
    ref2.load_state_dict(sd)
    st = ref2.state_dict()["state"]
    assert len(st) == len(m0) + len(e0) and int(st[0]["step"]) == 4
    for i, p in enumerate(m0 + e0):
        for k in ("exp_avg", "exp_avg_sq"):
            want = ref.state_dict()["state"][i][k]
            assert st[i][k].shape == want.shape and (st[i][k] - want).abs().max() <= 2e-6 * want.abs().max(), (i, k)
    m2, e2 = copy.deepcopy(m0), copy.deepcopy(e0)
    opt2 = FlatAdam([{"params": m2}, {"params": e2, "lr": 5.0}], lr=7.0, betas=(0.1, 0.2))
    opt2.load_state_dict(ref.state_dict())
    assert [g["step"] for g in opt2.param_groups] == [4, 4] and opt2.param_groups[1]["lr"] == 2.0e-3 and tuple(opt2.param_groups[0]["betas"]) == (0.5, 0.9)
    gs = _grads(m0 + e0, 99)
    ref.zero_grad()
    opt2.zero_grad()
    for p, q, g in zip(m0 + e0, m2 + e2, gs):
        p.grad = g.clone()
        q.grad.add_(g)
    ref.step()
    opt2.step()
    for p, q in zip(m0 + e0, m2 + e2):
        assert (p - q).abs().max() <= 2e-6 * p.abs().max()


def test_flat_adam_takes_over_a_gradient_that_replaced_its_view():
    from favae_step import FlatAdam
    m, e = _params(1)
    opt = FlatAdam(m + e, lr=1e-2, betas=(0.5, 0.9))
    before = [p.detach().clone() for p in m + e]
    for p in m + e:
        p.grad = None                                            # a loop that sets gradients to None itself
    loss = sum((p * p).sum() for p in m + e)
    loss.backward()                                              # autograd assigns fresh tensors
    opt.step()
    for p, b in zip(m + e, before):
        # first Adam step: |update| = lr for every element with a non-zero gradient (2 p here)
        assert torch.allclose((p - b).abs(), torch.full_like(b, 1e-2), rtol=1e-3), "the fresh gradients were used"
    opt.zero_grad()
    for p, view in zip(m + e, opt._flat[0][4]):
        assert p.grad is view and float(p.grad.abs().max()) == 0.0


def test_flat_adam_refuses_cpu_parameters():
    from favae_step import FlatAdam
    with pytest.raises(RuntimeError, match="no CPU path"):
        FlatAdam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)


def test_flat_adam_direct_mode_is_the_default_only_without_a_multi_rank_process_group():
    from favae_step import FlatAdam
    m, e = _params(2)
    assert FlatAdam(m + e, lr=1e-3).direct_grads is True          # one process: nothing to exchange, nothing hooks AccumulateGrad


@pytest.mark.parametrize("direct", [False, True])
def test_flat_adam_on_the_drop_in_modules(direct):
    """two ResnetBlocks (fused GroupNorm + SiLU + conv kernels, weight gradients on the second stream): the gradients that reach the flat
    buffer -- through AccumulateGrad (direct_grads=False, the DDP-safe mode) or accumulated by the kernels themselves (True) -- are those
    of an ordinary backward pass, and one step moves the parameters as torch.optim.Adam moves a copy"""
    from favae_step import FlatAdam
    from models import codec as C
    torch.manual_seed(0)
    net = torch.nn.Sequential(C.ResnetBlock(64, 128, 0.0), C.ResnetBlock(128, 128, 0.0)).to(DEV)
    ref_net = copy.deepcopy(net)
    x = torch.randn(2, 64, 32, 32, device=DEV)
    gy = torch.randn(2, 128, 32, 32, device=DEV)
    ref_opt = torch.optim.Adam(ref_net.parameters(), lr=1e-3, betas=(0.5, 0.9))
    ref_net(x).backward(gy)
    want = [p.grad.detach().clone() for p in ref_net.parameters()]
    ref_opt.step()
    opt = FlatAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.9), direct_grads=direct)
    assert opt.direct_grads is direct and all(getattr(p, "_favae_flat", None) is direct for p in net.parameters())
    for it in range(2):                                          # twice: zero_grad() between them must leave nothing behind
        opt.zero_grad()
        net(x).backward(gy)
        for p, w in zip(net.parameters(), want):
            assert (p.grad - w).abs().max() <= 1e-5 * w.abs().max() + 1e-7
    opt.step()
    for p, q in zip(net.parameters(), ref_net.parameters()):
        assert (p - q).abs().max() <= 1e-5 * q.abs().max() + 2e-6


def test_autograd_grad_inside_a_flat_adam_loop_leaves_the_flat_gradients_alone():
    """the adaptive weight of the GAN stage (favae_scripts/train_favae.py:32-39) is a torch.autograd.grad() w.r.t. a conv weight in the
    middle of the user's loop: with FlatAdam's direct accumulation on, that call must get a real gradient tensor and must not add
    anything to the flat gradient buffer; the .backward() that follows accumulates as usual"""
    from favae_step import FlatAdam
    from models import codec as C
    torch.manual_seed(0)
    net = torch.nn.Sequential(C.ResnetBlock(64, 128, 0.0), C.ResnetBlock(128, 128, 0.0)).to(DEV)
    ref_net = copy.deepcopy(net)
    x = torch.randn(2, 64, 32, 32, device=DEV)
    gy = torch.randn(2, 128, 32, 32, device=DEV)
    w_ref = ref_net[1].block[6].weight
    want_w = torch.autograd.grad(ref_net(x), w_ref, gy)[0]
    ref_net(x).backward(gy)
    want = [p.grad.detach().clone() for p in ref_net.parameters()]
    opt = FlatAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.9), direct_grads=True)
    opt.zero_grad()
    out = net(x)
    got_w = torch.autograd.grad(out, net[1].block[6].weight, gy, retain_graph=True)[0]
    assert got_w is not None and (got_w - want_w).abs().max() <= 1e-5 * want_w.abs().max()
    torch.cuda.synchronize()
    assert float(opt._flat[0][1].abs().max()) == 0.0, "autograd.grad() must not touch .grad"
    out.backward(gy)
    for p, w in zip(net.parameters(), want):
        assert p.grad is not None and (p.grad - w).abs().max() <= 1e-5 * w.abs().max() + 1e-7
    # backward(inputs=[...]) accumulates into exactly those parameters
    opt.zero_grad()
    net(x).backward(gy, inputs=[net[0].block[2].weight])
    torch.cuda.synchronize()
    for (name, p), w in zip(net.named_parameters(), want):
        if name == "0.block.2.weight":
            assert (p.grad - w).abs().max() <= 1e-5 * w.abs().max()
        else:
            assert float(p.grad.abs().max()) == 0.0, name
