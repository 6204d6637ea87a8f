"""Worker: the mirrored reference unit tests (tests/test_reference_suite.py) on
every rank of a torchrun job, as the reference runs its own suite under mpirun.
All ranks must run the same tests in the same order (they communicate)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
rc = pytest.main(['-x', '-q', '-p', 'no:cacheprovider',
                  os.path.join(HERE, 'test_reference_suite.py')])
if rc != 0:
    sys.exit(int(rc))
if int(os.environ.get('RANK', '0')) == 0:
    print('mp_refsuite_worker ok')
