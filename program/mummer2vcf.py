#!/usr/bin/env python3
"""Drop-in for Quasimodo's program/mummer2vcf.py (same options), without Biopython."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quasimodo_amd.mummer2vcf import main

if __name__ == "__main__":
    sys.exit(main())
