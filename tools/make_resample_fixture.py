#!/usr/bin/env python3
"""tests/golden/lj001_resample.npz: the ONE fixture the reference itself holds for the sample-rate conversion in front of the path.

/root/reference/I_ea/hifi_gan/test_files/LJ001-0001_22k.wav and LJ001-0001_16k.wav are the same utterance at 22.05 kHz and 16 kHz
(int16 PCM).  This script copies an excerpt of each -- DATA, the files' samples -- so that the oracle's restatement of librosa
0.9.1's `kaiser_best` resampler (resampy; absent from the image) and the GPU kernel can be pinned where the reference cannot travel:
3 s of the 22.05 kHz file starting at a multiple of 441 samples (an exact multiple of 320 output samples) and the 16 kHz samples
the interior of that excerpt maps to.  Authoring container only.

usage: python tools/make_resample_fixture.py [--out tests/golden]
"""
import argparse
import json
import os

import numpy as np
from scipy.io import wavfile

REF = "/root/reference/I_ea/hifi_gan/test_files"

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    a = ap.parse_args()
    sr22, x22 = wavfile.read(os.path.join(REF, "LJ001-0001_22k.wav"))
    sr16, x16 = wavfile.read(os.path.join(REF, "LJ001-0001_16k.wav"))
    assert (sr22, sr16) == (22050, 16000) and x22.dtype == np.int16 and x16.dtype == np.int16
    k0, k1 = 40, 130                                   # excerpt [k0 * 441, k1 * 441) = 0.8 s .. 2.6 s of the 22.05 kHz file
    seg22 = x22[k0 * 441: k1 * 441]
    seg16 = x16[k0 * 320: k1 * 320]                    # the output samples of the same time span
    kt = (x22.shape[0] - 4000) // 441                  # the file's END from a whole multiple of 441 input samples on
    n_res = int(x22.shape[0] * 16000 / 22050)          # resampy's output length; the 16 kHz file is zero beyond it
    meta = dict(source="I_ea/hifi_gan/test_files/LJ001-0001_{22k,16k}.wav", start22=k0 * 441, start16=k0 * 320, n22=int(x22.shape[0]),
                n16=int(x16.shape[0]), tail_start22=kt * 441, tail_start16=kt * 320, n_resampled=n_res,
                zeros_after=bool((x16[n_res:] == 0).all()))
    np.savez_compressed(os.path.join(a.out, "lj001_resample.npz"), seg22=seg22, seg16=seg16, head22=x22[:9 * 441], head16=x16[:9 * 320],
                        tail22=x22[kt * 441:], tail16=x16[kt * 320: n_res + 8], meta=np.array(json.dumps(meta)))
    print("wrote", os.path.join(a.out, "lj001_resample.npz"), seg22.shape, seg16.shape, meta)
