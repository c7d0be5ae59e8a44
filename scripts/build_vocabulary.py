#!/usr/bin/env python3
"""Build the caption vocabulary files - counterpart of the reference's var_updown/scripts/build_vocabulary.py (same flags)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "style-seqcvae_amd"))
from ssc_runtime.vocab_builder import build_from_files  # noqa: E402

parser = argparse.ArgumentParser(description="Build a vocabulary out of COCO train2017 (+ SentiCap) captions json files.")
parser.add_argument("-c", "--captions-jsonpath", default="data/coco/captions_train2017.json")
parser.add_argument("-t", "--word-count-threshold", type=int, default=5)
parser.add_argument("-o", "--output-dirpath", default="data/vocabulary")
parser.add_argument("-s", "--senticap-jsonpath", default="data/SentiCap/data/senticap_dataset.json")
parser.add_argument("-st", "--senticap-word-count-threshold", type=int, default=2)

if __name__ == "__main__":
    a = parser.parse_args()
    senti = a.senticap_jsonpath if os.path.exists(a.senticap_jsonpath) else None
    vocab = build_from_files(a.captions_jsonpath, senti, a.output_dirpath, a.word_count_threshold, a.senticap_word_count_threshold)
    print(f"Caption vocabulary size (with special tokens): {len(vocab)}")
