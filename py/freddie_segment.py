#!/usr/bin/env python3
"""Drop-in for the reference's py/freddie_segment.py: same command line, same files in and out.
The work is done by freddie_amd (gfx950 HIP library behind a C-ABI); see INTEGRATION.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from freddie_amd.segment import main  # noqa: E402

if __name__ == "__main__":
    main()
