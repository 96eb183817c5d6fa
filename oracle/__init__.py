"""TEST INFRASTRUCTURE ONLY — CPU oracle for the S3GRL PoS / PoS Plus / SoP operator precompute.

Nothing in the product package (`s3grl_amd/`) may import this package.  The only legal
importers are `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`,
and there only as the checker / the timed CPU baseline — never as a compute fallback.

Parity status: the *extraction* half (k-hop node sets, hop distances, masked induced sub-CSR,
common-neighbour sets) is pinned against the reference's own `utils.k_hop_subgraph` /
`utils.neighbors` executed in the build container (tests/golden/make_golden.py, fixtures in
tests/golden/extract_*.npz).  The *diffusion* half (normalisation, operator powers, SpMM)
lives in third-party packages absent from the reference tree and from this image
(torch-sparse==0.6.13, torch_geometric; pins: reference quick_install.sh:7-8); the reference
holds no tests or golden vectors for it, so that half is "parity unpinned" by the reference
and is anchored on hand-derived known answers (tests/test_oracle_known_answers.py) instead.
"""
from .s3grl_oracle import (  # noqa: F401
    neighbors,
    k_hop_subgraph,
    normalized_subgraph_operator,
    pos_link,
    get_PoS_prepped_ds,
    get_PoS_Plus_prepped_ds,
    global_normalized_powers,
    get_SoP_prepped_ds,
    get_SoP_restricted_ds,
    hybrid_combine,
    centre_pool,
    collate_rows,
    hash_sampler,
    hop_sample_key,
)
