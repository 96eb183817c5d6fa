"""s3grl_amd — MI355X (gfx950) engine for the S3GRL PoS / PoS Plus / SoP operator precompute.

Only the hot path of venomouscyanide/S3GRL lives here: the HIP kernels and
C ABI in csrc/ + include/s3grl.h, and the host-side mirror of the reference's operator
interface (`s3grl_amd.tuned_SIGN`).  Importing the package does not touch the GPU; the first
call into the engine loads libs3grl_hip.so and fails loudly if it is not built.
"""
__version__ = "0.1.0"


def precompute(*args, **kwargs):
    """See `s3grl_amd.engine.precompute` (imported lazily: the engine imports torch)."""
    from .engine import precompute as _p

    return _p(*args, **kwargs)


def extract_enclosing_subgraphs(*args, **kwargs):
    """See `s3grl_amd.dataset.extract_enclosing_subgraphs` (reference utils.py:446-554)."""
    from .dataset import extract_enclosing_subgraphs as _f

    return _f(*args, **kwargs)


def process_split(*args, **kwargs):
    """See `s3grl_amd.dataset.process_split` (reference sgrl_link_pred.py:96-220)."""
    from .dataset import process_split as _f

    return _f(*args, **kwargs)
