"""seekmer_amd: the `seekmer infer` hot path (k-mer pseudoalignment,
equivalence-class counting, EM / bootstrap) as hand-written HIP kernels for
MI355X (gfx950) behind a C ABI, with the reference's Python module surface
(`common`, `mapper`, `infer`, `index_builder`) on top.  See DESIGN.md."""
from .common import *          # noqa: F401,F403
from .mapper import *          # noqa: F401,F403

__version__ = '2020.0.0+mi355x.1'
