"""`pytorch_mppi_amd.mppi` -- the module path of the reference's `pytorch_mppi.mppi`
(/root/reference/src/pytorch_mppi/mppi.py), kept as the import location of the whole controller family; the code lives
in controller.py (MPPI), variants.py (SMPPI, KMPPI, MPPI_Batched, run_mppi), draws.py, forms.py and jit_glue.py."""
from ._util import SpecificActionSampler, _ptr  # noqa: F401
from .controller import MPPI, GraphedCommand  # noqa: F401
from .jit_glue import _auto_jit_mode  # noqa: F401
from .variants import SMPPI, KMPPI, MPPI_Batched, TimeKernel, RBFKernel, run_mppi  # noqa: F401
