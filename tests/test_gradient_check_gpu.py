"""GradientChecker-style tests (include/caffe/test/test_gradient_check_util.hpp: CheckGradientExhaustive with
stepsize 1e-2, threshold 1e-3 -- how the reference tests every layer that has a test, e.g.
test_contrastive_loss_layer.cpp:90-101, test_embed_layer.cpp:137-175) for the three layers of the path, which
the reference ships WITHOUT tests: the analytic Backward of Layer<double> on the GPU against central
differences of its own Forward, for every bottom element and every parameter element, objective
sum_i top_i * c_i with fixed random c (the checker's "top_id / top_data_id" loop folded into one objective)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEP, THRESHOLD = 1e-2, 1e-3


@pytest.fixture(scope="module")
def L(hiplib):
    from mms_answer_selection_amd import layers
    layers.lib()
    layers.set_mode_gpu()
    return layers


def check_gradient_exhaustive(L, proto, bottoms, params=(), check_bottom=None, kink=None):
    bottoms = [np.array(b, np.float64) for b in bottoms]
    params = [np.array(p, np.float64) for p in params]
    pd = None if check_bottom is None else [i in check_bottom for i in range(len(bottoms))]
    top0, _, _ = L.run_layer_f64(proto, bottoms, params=params, propagate_down=pd)
    c = np.random.default_rng(5).standard_normal(top0.shape)

    def objective(bs, ps):
        top, _, _ = L.run_layer_f64(proto, bs, params=ps, propagate_down=pd)
        return float((top * c).sum())

    _, bdiffs, pdiffs = L.run_layer_f64(proto, bottoms, top_diff=c, params=params,
                                        param_diffs=[np.zeros_like(p) for p in params], propagate_down=pd)
    checked = 0
    for kind, arrays, grads in (("bottom", bottoms, bdiffs), ("param", params, pdiffs)):
        for ai, (x, g) in enumerate(zip(arrays, grads)):
            if kind == "bottom" and check_bottom is not None and ai not in check_bottom:
                continue
            flat = x.reshape(-1)
            for i in range(flat.size):
                if kink is not None and kink(kind, ai, i):
                    continue
                keep = flat[i]
                flat[i] = keep + STEP
                fp = objective(bottoms, params)
                flat[i] = keep - STEP
                fm = objective(bottoms, params)
                flat[i] = keep
                est = (fp - fm) / (2 * STEP)
                got = g.reshape(-1)[i]
                scale = max(abs(got), abs(est), 1.0)         # test_gradient_check_util.hpp:150-160
                assert abs(got - est) <= THRESHOLD * scale, (kind, ai, i, got, est)
                checked += 1
    return checked


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_simcross_gradient(L, mode):
    r = np.random.default_rng(40 + mode)
    N, W1, W2, D, M = 2, 3, 2, 4, 2
    q = r.standard_normal((N, W1, D))
    a = r.standard_normal((N, W2, D))
    if mode == 2:
        proto = ('layer { name: "s" type: "SimCross" bottom: "q" bottom: "a" top: "t" '
                 'sim_cross_param { dist_mode: 2 mesure_count: %d bias_term: true } }' % M)
        params = [r.standard_normal((M, D, D)) * 0.5, r.standard_normal((M, W1, W2))]
    else:
        proto = ('layer { name: "s" type: "SimCross" bottom: "q" bottom: "a" top: "t" '
                 'sim_cross_param { dist_mode: %d } }' % mode)
        params = []
    n = check_gradient_exhaustive(L, proto, [q, a], params)
    assert n == q.size + a.size + sum(p.size for p in params)


def test_simmatrix_gradient(L):
    r = np.random.default_rng(50)
    N, K1, K2 = 3, 4, 5
    q, a = r.standard_normal((N, K1)), r.standard_normal((N, K2))
    W = r.standard_normal((K1, K2)) * 0.5
    proto = 'layer { name: "m" type: "SimMatrix" bottom: "q" bottom: "a" top: "t" }'
    assert check_gradient_exhaustive(L, proto, [q, a], [W]) == q.size + a.size + W.size


def test_pairrankloss_gradient(L):
    r = np.random.default_rng(60)
    N = 12
    a, b = r.uniform(0, 1, (N, 1)), r.uniform(0, 1, (N, 1))
    y = (r.uniform(size=(N, 1)) < 0.5).astype(np.float64)
    margin = float(np.float32(0.3))
    # keep every pair at least 2 steps away from the hinge's and the |.|'s kinks (the checker's `kink` argument)
    d = a - b
    near = (np.abs(margin - d * y) < 3 * STEP) | (np.abs(d) < 3 * STEP)
    a[near] += 0.2
    proto = ('layer { name: "l" type: "PairRankLoss" bottom: "a" bottom: "b" bottom: "y" top: "loss" '
             'pair_rank_loss_param { margin: 0.3 } }')
    # labels are not differentiated: propagate_down[2] must be false (LOG(FATAL) otherwise, :57-60)
    assert check_gradient_exhaustive(L, proto, [a, b, y], check_bottom={0, 1}) == 2 * N
