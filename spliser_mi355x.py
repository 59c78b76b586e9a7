#!/usr/bin/env python3
"""Drop-in entry script: same sub-commands and flags as SpliSER_v0_1_8.py (see spliser_amd/cli.py)."""
import sys

from spliser_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
