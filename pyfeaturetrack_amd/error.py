"""KLTError / KLTWarning (reference: error.py:12-28).

The reference's KLTError prints the message and calls exit(1).  The same name is kept for
API compatibility; it still terminates (by raising SystemExit, which is what exit(1) does)
after printing.  Nothing beneath the C ABI ever exits the process.
"""
from __future__ import print_function


def KLTError(err):
    print(err)
    raise SystemExit(1)


def KLTWarning(err):
    print(err)
