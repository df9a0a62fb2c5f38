#!/usr/bin/env python3
"""Entry point with the reference's name and CLI: ``python predict.py config=unet config.ckpt=/abs/path``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mi355seg  # noqa: E402
from mi355seg.predict import main  # noqa: E402

if __name__ == "__main__":
    main()
