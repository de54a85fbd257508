#!/usr/bin/env python3
"""Drop-in for `python predict.py` of the reference's I_ea directory: reads ./predict.yaml (see speech_inpainting_amd/predict.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from speech_inpainting_amd.predict import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
