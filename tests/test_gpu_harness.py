"""End to end: engine output -> SIGNNet twin with the HIP pooling -> AUC.
BASELINE config 1 shape: USAir, PoS / PoS Plus, 1-hop; features are a synthetic stand-in for the
paper's node2vec vectors, so the check is functional (the link signal is learnt), not a number
from the paper."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,k_heuristic,strategy", [("pos", 0, ""), ("pos_plus", 1, "mean")])
def test_usair_end_to_end_auc(mode, k_heuristic, strategy):
    import torch
    from s3grl_amd import workloads
    from s3grl_amd.engine import Engine
    from s3grl_amd.harness import train_and_evaluate

    w = workloads.make("usair_pos_k2")
    eng = Engine("cuda:0")
    G = eng.graph(w.A)
    f = eng.features(w.X)

    def prep(split):
        pos, neg = w.split.links[split]
        li = np.concatenate([pos, neg], axis=1)
        y = torch.cat([torch.ones(pos.shape[1]), torch.zeros(neg.shape[1])]).to(eng.device)
        res = eng.precompute(G, f, eng.links(li), mode=mode, num_hops=1, sign_k=2)
        return res.rows, res.row_ptr, y

    auc, _ = train_and_evaluate(prep("train"), prep("test"), k_heuristic=k_heuristic,
                                k_pool_strategy=strategy, epochs=8, lr=2e-3, seed=1)
    assert auc > 0.85, auc
    eng.close()
