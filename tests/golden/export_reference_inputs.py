#!/usr/bin/env python3
"""Export the LLRs of tests/golden/nmsq_golden.npz as tests/golden/reference_inputs.mat, the input of
matlab/dump_reference_vectors.m: the ONE route by which the decoder oracle can ever be pinned to the reference's own
arithmetic (comm.LDPCDecoder is closed source and no MATLAB exists in the build image -- SURVEY.md section 8c).

On a machine with MATLAB + Communications Toolbox and a checkout of robmaunder/ldpc-3gpp-matlab:
    >> addpath('<checkout>'); cd <repo>/matlab; dump_reference_vectors
writes tests/golden/reference_outputs.mat (hard decisions + NumIterations of comm.LDPCDecoder on these LLRs);
tests/test_reference_dump.py then compares oracle/orc_decode_bp_flood with it (and is skipped while the file is absent).

One struct array `cases` with fields name, BG, Z, iterations (row vector of caps), llr ((ncols*Z) x batch double, MATLAB
column-major: one codeword per column, exactly the cw_tilde of NRLDPCDecoder.m:262-265 -- 0 for the punctured / untransmitted
positions).
Run from the repo root:  python tests/golden/export_reference_inputs.py
"""
import os

import numpy as np
from scipy.io import savemat

HERE = os.path.dirname(os.path.abspath(__file__))
# the reference's own iteration caps: the script default (plot_BLER_vs_SNR.m:37) and the object default (NRLDPCDecoder.m:41)
ITERATIONS = (8, 50)


def cases():
    d = np.load(os.path.join(HERE, "nmsq_golden.npz"))
    names = sorted({k.split("/")[0] for k in d.files})
    out = []
    for n in names:
        bg, Z = int(d[n + "/cfg"][0]), int(d[n + "/cfg"][1])
        llr = d[n + "/llr"].astype(np.float64)  # [batch][ncols*Z]
        out.append({"name": n, "BG": float(bg), "Z": float(Z), "iterations": np.array(ITERATIONS, np.float64)[None, :], "llr": llr.T.copy()})
    return out


def main():
    cs = cases()
    arr = np.zeros(len(cs), dtype=[("name", "O"), ("BG", "O"), ("Z", "O"), ("iterations", "O"), ("llr", "O")])
    for i, c in enumerate(cs):
        for k in c:
            arr[i][k] = c[k]
    path = os.path.join(HERE, "reference_inputs.mat")
    savemat(path, {"cases": arr}, do_compression=True, oned_as="column")
    print("wrote", path, os.path.getsize(path), "bytes,", len(cs), "cases")


if __name__ == "__main__":
    main()
