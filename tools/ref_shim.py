"""Dev-container-only shim that makes the read-only reference importable.

It refers to the reference by path (``/root/reference``) and copies nothing
from it.  It is used ONLY by ``tools/gen_golden.py`` to generate the golden
vectors committed under ``tests/golden/`` and is never imported by the
package, the tests, ``bench.py`` or ``__graft_entry__`` (the reference does
not exist on the GPU box).

Why a shim: ``import tfep`` raises ModuleNotFoundError for the generated
``tfep._version`` module and ``tfep.utils.misc`` imports ``pint`` (absent
here, unused on the flow path).  Both are ordinary import errors, worked
around by pre-seeding ``sys.modules``.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('TFEP_REFERENCE_ROOT', '/root/reference')


def install():
    if 'tfep' in sys.modules:
        return
    sys.dont_write_bytecode = True
    pkg = types.ModuleType('tfep')
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'tfep')]
    sys.modules['tfep'] = pkg
    if 'pint' not in sys.modules:
        pint = types.ModuleType('pint')
        pint.errors = types.SimpleNamespace(DimensionalityError=Exception)
        pint.Quantity = object
        pint.Unit = object
        sys.modules['pint'] = pint


install()
