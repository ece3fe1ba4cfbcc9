"""NeuralPointsRayMarching.prefetch_query: the next batch's query one step ahead on a side stream.  Same kernels, same jitter seed sequence:
the step that consumes a prefetched query must give bit-identical outputs and gradients; a prefetched result for a different batch is dropped."""
import pytest
import torch

import bench
from pointnerf_amd import config

pytestmark = pytest.mark.gpu


def _run(model, opt, inputs, prefetch):
    npnt, agg = model.neural_points, model.aggregator
    params = list(agg.parameters()) + [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
    npnt.querier.count = 100                    # the jitter seed of a query follows the querier's call counter: same sequence for both runs
    outs = []
    for i, inp in enumerate(inputs):
        for p in params:
            p.grad = None
        if prefetch and i + 1 < len(inputs):
            pass
        out = model(**inp)
        if prefetch and i + 1 < len(inputs):
            assert model.prefetch_query(**inputs[i + 1])          # issued while this step's loss / backward are still to come
        loss = bench.loss_fn(opt, out, inp, 1)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((out["coarse_raycolor"].clone(), out["ray_mask"].clone(), float(loss), [p.grad.clone() for p in params]))
    return outs


def test_prefetched_query_gives_identical_steps():
    dev = torch.device("cuda:0")
    opt = config.bench_lego_opt(is_train=1)
    model = bench.build_model(opt, 200_000, dev)
    inputs = [bench.step_inputs(i, 0, 1, 2048, dev) for i in range(3)]
    with torch.no_grad():
        model(**inputs[0])                       # builds the grid (the first query always runs on the main stream)
    a = _run(model, opt, inputs, prefetch=False)
    b = _run(model, opt, inputs, prefetch=True)
    for (ca, ma, la, ga), (cb, mb, lb, gb) in zip(a, b):
        assert torch.equal(ma, mb) and torch.equal(ca, cb) and la == lb
        for x, y in zip(ga[:2], gb[:2]):         # MLP gradients: split-K partial sums in a fixed order -> identical
            assert torch.equal(x, y)
        for x, y in zip(ga[-4:], gb[-4:]):       # point gradients are sums of atomics: equal up to their order
            assert float((x - y).abs().max()) <= 1e-6 * max(float(x.abs().max()), 1e-12)
    # a prefetched result for another batch is dropped, not used
    assert model.prefetch_query(**inputs[2])
    npnt = model.neural_points
    npnt.querier.count = 100
    with torch.no_grad():
        o = model(**inputs[0])
    assert model._prefetched is None and torch.equal(o["ray_mask"], a[0][1])
