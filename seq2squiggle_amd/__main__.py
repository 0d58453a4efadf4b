"""`python -m seq2squiggle_amd ...`.  When the command has loaded torch (i.e. it predicted), the process leaves through os._exit once
click is done: every output file is closed by then, and the interpreter's and torch's teardown (0.3 - 0.5 s with a live HIP context)
buys a command-line run nothing."""
import logging
import os
import sys

from .cli import main

try:
    main()
except SystemExit as e:
    code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    if e.code is not None and not isinstance(e.code, int):
        print(e.code, file=sys.stderr)
    if "torch" not in sys.modules:
        raise
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)
