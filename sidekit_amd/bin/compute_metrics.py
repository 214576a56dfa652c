"""EER from a score file and a key file -- the in-tree part of ``tools/compute_metrics.py``.

The reference prints EER, Cllr and linkability using ``anonymization_metrics.performance`` (cloned at
install time, ``install.sh:98-105``, not in the tree): only the EER is built, from the in-tree
``rocch`` / ``rocch2eer``.  Files: ``enrol test score`` and ``enrol test target|nontarget`` per line.
"""
import argparse

import numpy

from ..bosaris import Key, Scores, rocch, rocch2eer


def eer_from_files(score_file, key_file):
    scores = Scores.read_txt(score_file)
    key = Key.read_txt(key_file)
    tar, non = scores.get_tar_non(key)
    return rocch2eer(*rocch(numpy.asarray(tar, dtype=numpy.float64), numpy.asarray(non, dtype=numpy.float64)))


def cli(argv=None):
    parser = argparse.ArgumentParser(description='EER of a score file against a key file')
    parser.add_argument('-s', dest='score_file', type=str, required=True)
    parser.add_argument('-k', dest='key_file', type=str, required=True)
    args = parser.parse_args(argv)
    print("EER: {:.2f}".format(eer_from_files(args.score_file, args.key_file) * 100))


if __name__ == '__main__':
    cli()
