"""Trial list + two x-vector scp files -> ``enrol test score`` lines: ``sidekit/bin/compute_spk_cosine.py``.

Semantics kept (``:29-67``): enrolment x-vectors are averaged per speaker (``enroll_utt2spk``) and
L2-normalised, every listed trial ``<enrol speaker> <test utterance>`` gets ``1 - cosine distance``.
The per-trial python loop (``:18-26``) becomes one launch of ``sc_cosine_trials`` (float64 maths on
float32 vectors, like scipy on the reference's float32 arrays).
"""
import argparse
import ctypes
import os

import numpy
import torch

from .. import _lib
from ..kaldi_io import read_scp


def read_utt2spk_file(utt2spk_file):
    utt2spk = {}
    with open(utt2spk_file) as f:
        for line in f:
            parts = line.strip().split()
            utt2spk[parts[0]] = parts[1]
    return utt2spk


def listed_trial_scores(enroll_matrix, test_matrix, enroll_idx, test_idx, device="cuda"):
    """``out[k] = cos(enroll_matrix[enroll_idx[k]], test_matrix[test_idx[k]])`` on the GPU (float64 result)."""
    if not torch.cuda.is_available():
        raise RuntimeError("sidekit_amd computes on the GPU only (no CPU fallback) and no GPU is visible")
    device = torch.device(device)
    E = torch.as_tensor(numpy.ascontiguousarray(enroll_matrix, dtype=numpy.float32)).to(device)
    T = torch.as_tensor(numpy.ascontiguousarray(test_matrix, dtype=numpy.float32)).to(device)
    ei = torch.as_tensor(numpy.ascontiguousarray(enroll_idx, dtype=numpy.int32)).to(device)
    ti = torch.as_tensor(numpy.ascontiguousarray(test_idx, dtype=numpy.int32)).to(device)
    out = torch.empty(ei.numel(), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().sc_cosine_trials(E.data_ptr(), T.data_ptr(), E.shape[1], ei.data_ptr(), ti.data_ptr(), ei.numel(),
                                               out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return out.cpu().numpy()


def main(args):
    trials = [x.split() for x in open(args.trials)]
    utt1s = [x[0] for x in trials]
    utt2s = [x[1] for x in trials]
    utt2embd_enroll = {u: e.reshape(-1) for u, e in read_scp(args.enroll_scp)}
    utt2embd_trial = {u: e.reshape(-1) for u, e in read_scp(args.trial_scp)}
    utt2spk = read_utt2spk_file(args.enroll_utt2spk)
    spk2utt = {}
    for utt, spk in utt2spk.items():
        spk2utt.setdefault(spk, []).append(utt)
    spk_ids = list(spk2utt)
    spk_mean = numpy.stack([numpy.mean([utt2embd_enroll[u] for u in spk2utt[s]], axis=0) for s in spk_ids]).astype(numpy.float32)
    spk_mean /= numpy.linalg.norm(spk_mean, ord=2, axis=1, keepdims=True)
    tst_ids = list(utt2embd_trial)
    tst = numpy.stack([utt2embd_trial[u] for u in tst_ids])
    spk_row = {s: i for i, s in enumerate(spk_ids)}
    tst_row = {u: i for i, u in enumerate(tst_ids)}
    scores = listed_trial_scores(spk_mean, tst, [spk_row[u] for u in utt1s], [tst_row[u] for u in utt2s], getattr(args, "device", "cuda"))
    with open(args.output, "w") as f:
        for enroll, trial, score in zip(utt1s, utt2s, scores):
            f.write(" ".join([enroll, trial, str(score)]) + "\n")


def cli(argv=None):
    parser = argparse.ArgumentParser('Speaker Verification Trials/Enroll Cosine Calculation.')
    parser.add_argument('trials')
    parser.add_argument('enroll_utt2spk')
    parser.add_argument('trial_scp')
    parser.add_argument('enroll_scp')
    parser.add_argument('output')
    parser.add_argument('--device', default="cuda")
    args = parser.parse_args(argv)
    for p in (args.trials, args.enroll_utt2spk, args.enroll_scp, args.trial_scp):
        assert os.path.isfile(p), "NO SUCH FILE: %s" % p
    main(args)


if __name__ == '__main__':
    cli()
