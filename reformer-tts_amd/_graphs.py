"""hipGraph capture helper shared by the trainer and the inference paths."""
from __future__ import annotations

import contextlib
import gc

import torch


@contextlib.contextmanager
def capturing(graph, **kw):
    """``torch.cuda.graph`` with the Python garbage collector held off: a collection that happens to run inside a
    capture can release device resources (events, blocks that were used on another stream), which HIP refuses while
    a stream is capturing -- the process then aborts from a destructor.  Seen for real: ``Tensor.backward(grad)``
    lazily imports ``torch.fx.experimental.symbolic_shapes`` (sympy), whose thousands of allocations trigger a
    collection in the middle of a capture.  Both causes are removed: the import happens up front, and automatic
    collection is off for the duration of the capture."""
    import torch.fx.experimental.symbolic_shapes  # noqa: F401  (what autograd imports on first use of grad_tensors)
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_enabled:
            gc.enable()
