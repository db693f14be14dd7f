"""N>1 path on CPU: world_size-2 gloo processes.  (a) batch-sharded inference reproduces the single-process logits
exactly and in order; (b) bucketed gradient all-reduce reproduces the single-process gradients of the global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from peekvit_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from peekvit_amd import dist
    from peekvit_amd.models.vit import VisionTransformer
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    model = VisionTransformer(**cfg)
    synth.load_synth_weights(model, cfg)
    x = torch.from_numpy(synth.synth_images(5, cfg["image_size"]))          # odd batch: ragged shards (3 + 2)
    logits = dist.sharded_forward(model.eval(), x)
    # one training step's gradients: local loss is the SUM over the shard / global batch, so sum-reduce = global mean
    model.train()
    y = torch.arange(5) % cfg["num_classes"]
    out = model(dist.shard_batch(x))
    loss = torch.nn.functional.cross_entropy(out, dist.shard_batch(y), reduction="sum") / x.shape[0]
    loss.backward()
    nb = dist.allreduce_gradients(model.parameters(), bucket_bytes=64 << 10, average=False)
    reduced = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # the same step again with the reduction OVERLAPPED with backward (hooks launch buckets while gradients still arrive)
    for p in model.parameters():
        p.grad = None
    reducer = dist.OverlappedGradReducer(model.parameters(), bucket_bytes=64 << 10, average=False)
    out = model(dist.shard_batch(x))
    (torch.nn.functional.cross_entropy(out, dist.shard_batch(y), reduction="sum") / x.shape[0]).backward()
    launched_during_backward = reducer.buckets_launched
    nb2 = reducer.finish()
    same = all(torch.equal(reduced[n], p.grad) for n, p in model.named_parameters() if p.grad is not None)
    # a third time through reducer.zero_grad(): the parameters' gradients ARE slices of the flat buckets (no per-step copy in, none back),
    # autograd accumulates into them and the all-reduce runs in place - same bits again
    reducer.zero_grad()
    views = all(p.grad.data_ptr() == v.data_ptr() for b in reducer._buckets for p, v in zip(b["params"], b["views"]))
    out = model(dist.shard_batch(x))
    (torch.nn.functional.cross_entropy(out, dist.shard_batch(y), reduction="sum") / x.shape[0]).backward()
    reducer.finish()
    views = views and all(p.grad.data_ptr() == v.data_ptr() for b in reducer._buckets for p, v in zip(b["params"], b["views"]))
    same = same and views and all(torch.equal(reduced[n], p.grad) for n, p in model.named_parameters() if p.grad is not None)
    # gradient ACCUMULATION (round-4 review): two micro-batches per step - the first backward inside no_sync() launches nothing and adds into
    # the bucket views, the second launches; the reduced result is the global gradient of both micro-batches.  A second backward OUTSIDE
    # no_sync() before finish() raises instead of racing the in-flight all-reduce.
    reducer.zero_grad()
    xs, ys = dist.shard_batch(x), dist.shard_batch(y)
    with reducer.no_sync():
        (torch.nn.functional.cross_entropy(model(xs[:1]), ys[:1], reduction="sum") / x.shape[0]).backward()
        launched_in_no_sync = reducer.buckets_launched
    (torch.nn.functional.cross_entropy(model(xs[1:]), ys[1:], reduction="sum") / x.shape[0]).backward()
    reducer.finish()
    accum = all(torch.allclose(reduced[n], p.grad, rtol=1e-4, atol=1e-6) for n, p in model.named_parameters() if p.grad is not None) and launched_in_no_sync == 0
    reducer.zero_grad()
    (torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum") / x.shape[0]).backward()
    raised = False
    try:
        (torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum") / x.shape[0]).backward()
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    reducer.finish()
    # every rank launches every bucket, gradient or not (a bucket without one travels as zeros: the collective sequence must not depend on which
    # parameters a rank's batch touched): a finish() with no backward at all still reduces all of them
    reducer.zero_grad()
    nb_uniform = reducer.finish()
    reducer.remove()
    # round 5: with `model=` the reducer all-reduces the fp16 training arithmetic's overflow verdict - a step dropped on ONE rank is dropped on all
    from peekvit_amd import train_engine
    red2 = dist.OverlappedGradReducer(model.parameters(), bucket_bytes=64 << 10, average=False, model=model)
    red2.zero_grad()
    (torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum") / x.shape[0]).backward()
    red2.finish()
    skip_clean = red2.skip_step
    red2.zero_grad()
    (torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum") / x.shape[0]).backward()
    train_engine.train_state(model).last_skipped = rank == 1          # as if rank 1's gradients had overflowed
    red2.finish()
    skip_one = red2.skip_step
    train_engine.train_state(model).last_skipped = False
    # round 6: the verdict exchange must not depend on rank-local state.  Rank 1's fp16 forward "overflowed": its operand goes to
    # bf16 for good - round 5 then left the collective out on that rank only and the other one blocked in it.  Now every rank takes part in every
    # finish(), and the sticky bf16 verdict reaches all of them together.
    st = train_engine.train_state(model)
    if rank == 1:
        st.operand = "bf16"
    red2.zero_grad()
    (torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum") / x.shape[0]).backward()
    red2.finish()                                                      # (a hang here is the bug)
    operand_after = st.operand
    skip_diverged = red2.skip_step
    # ... and a non-finite value in the REDUCED buckets (from whatever source) is a skip on every rank
    red2.zero_grad()
    (torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum") / x.shape[0]).backward()
    if rank == 0:
        with torch.no_grad():
            red2._buckets[0]["flat"][0] = float("inf")
    skipped0 = st.skipped
    red2.finish()
    skip_nonfinite, booked = red2.skip_step, st.skipped - skipped0
    st.operand, st.last_skipped = None, False
    red2.remove()
    if rank == 0:
        np.savez(os.path.join(out_dir, "r0.npz"), logits=logits.numpy(), nb=nb, nb2=nb2, early=launched_during_backward, same=same, accum=accum, raised=raised,
                 nb_uniform=nb_uniform, n_buckets=len(reducer._buckets), skip_clean=skip_clean, skip_one=skip_one,
                 operand_after=operand_after, skip_diverged=skip_diverged, skip_nonfinite=skip_nonfinite, booked=booked,
                 **{"g_" + n: g.numpy() for n, g in reduced.items()})
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "r0.npz"))
    from peekvit_amd.models.vit import VisionTransformer
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    model = VisionTransformer(**cfg)
    synth.load_synth_weights(model, cfg)
    x = torch.from_numpy(synth.synth_images(5, cfg["image_size"]))
    with torch.no_grad():
        ref = model.eval()(x).numpy()
    assert np.allclose(got["logits"], ref, rtol=0, atol=1e-6)            # same images, same order
    model.train()
    y = torch.arange(5) % cfg["num_classes"]
    torch.nn.functional.cross_entropy(model(x), y).backward()
    assert int(got["nb"]) > 1                                             # really bucketed
    assert int(got["nb2"]) > 1 and int(got["early"]) >= 1                 # overlapped reducer: buckets left before backward ended
    assert bool(got["same"])                                              # ... and produced bit-identical reduced gradients
    assert bool(got["accum"]) and bool(got["raised"])                     # no_sync() accumulation; a second backward outside it is refused
    assert int(got["nb_uniform"]) == int(got["n_buckets"])                # every bucket leaves on every rank
    assert not bool(got["skip_clean"]) and bool(got["skip_one"])          # rank 0 learns that rank 1 dropped its step
    # round 6: ranks whose operands diverged still meet in finish(); the sticky bf16 verdict of rank 1 reaches rank 0; a non-finite reduced bucket skips
    assert str(got["operand_after"]) == "bf16" and not bool(got["skip_diverged"])
    assert bool(got["skip_nonfinite"]) and int(got["booked"]) == 1
    for n, p in model.named_parameters():
        assert np.allclose(got["g_" + n], p.grad.numpy(), rtol=1e-4, atol=1e-6), n
