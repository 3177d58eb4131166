#!/usr/bin/env python3
"""Launcher of the trainer (protgen-gan_amd/train_cli.py): the package directory carries a hyphen, so it is
registered under an importable name first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pgload  # noqa: E402

pgload.load()
from protgen_gan_amd.train_cli import main  # noqa: E402

if __name__ == "__main__":
    main()
