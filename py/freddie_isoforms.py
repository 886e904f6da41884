#!/usr/bin/env python3
"""Drop-in for the reference's py/freddie_isoforms.py: same command line, same files in, same GTF out.
The per-read loops run on the GPU (freddie_amd/isoforms.py, include/freddie_isoforms.h); see INTEGRATION.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from freddie_amd.isoforms import main  # noqa: E402

if __name__ == "__main__":
    main()
