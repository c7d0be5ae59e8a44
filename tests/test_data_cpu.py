"""CPU: the batch producer's host logic (ssc_runtime/data.py) and the vocabulary builder (ssc_runtime/vocab_builder.py).
Reference semantics: updown-baseline/updown/data/datasets.py:150-202,623-632 (collate), var_updown/scripts/
build_vocabulary.py:55-136 (word selection, file format)."""
import json
import os

import numpy as np
import torch

from ssc_runtime import data as D
from ssc_runtime.vocab import Vocabulary
from ssc_runtime.vocab_builder import build_caption_vocabulary, build_from_files, caption_words, simple_tokenize


def test_collate_image_features_zero_pads_to_the_largest_instance():
    g = np.random.default_rng(0)
    xs = [g.standard_normal((n, 6)).astype(np.float32) for n in (3, 5, 1)]
    out = D.collate_image_features(xs)
    assert out.shape == (3, 5, 6) and out.dtype == np.float32
    for i, x in enumerate(xs):
        assert np.array_equal(out[i, :x.shape[0]], x) and not out[i, x.shape[0]:].any()
    # in-place form over a dirty, wider staging buffer
    buf = np.full((3, 8, 6), 7.0, dtype=np.float32)
    D.collate_image_features(xs, out=buf)
    assert np.array_equal(buf[:, :5], out) and not buf[0, 3:].any() and not buf[2, 1:].any()


def test_collate_captions_cuts_and_pads_with_id_zero():
    out = D.collate_captions([[5, 6, 7], [], list(range(2, 12))], 6)
    assert out.dtype == np.int64 and out.tolist() == [[5, 6, 7, 0, 0, 0], [0] * 6, [2, 3, 4, 5, 6, 7]]
    v = Vocabulary(["@@UNKNOWN@@", "@@BOUNDARY@@", "a", "dog"])
    assert D.tokens_to_ids(v, ["a", "zebra", "dog"]) == [2, 0, 3]          # out-of-vocabulary -> @@UNKNOWN@@ = padding id


def test_batch_order_is_a_function_of_seed_and_batch_number():
    n, gb = 50, 8
    a = [D.batch_indices(n, gb, 3, k) for k in range(14)]
    b = [D.batch_indices(n, gb, 3, k) for k in range(14)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    per_epoch = n // gb
    first = torch.cat(a[:per_epoch])
    assert first.unique().numel() == per_epoch * gb                     # no row twice within an epoch
    assert not torch.equal(torch.cat(a[per_epoch:2 * per_epoch]), first)   # reshuffled next epoch
    assert not torch.equal(D.batch_indices(n, gb, 4, 0), a[0])


def test_ragged_tensor_file_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(1)
    nb = torch.tensor([3, 1, 4, 2])
    feats = torch.randn(int(nb.sum()), 5, generator=g)
    torch.save({"features": feats, "num_boxes": nb, "caption_tokens": torch.randint(0, 9, (4, 6), generator=g),
                "sentiment": torch.tensor([[1.], [0.], [-1.], [0.]])}, tmp_path / "r.pt")
    d = D.TensorFileData(str(tmp_path / "r.pt"))
    assert len(d) == 4 and d.max_boxes() == 4 and d.feature_size() == 5 and d.ragged is not None


def test_vocabulary_builder_selection_and_files(tmp_path):
    coco = [{"id": i, "image_id": 100 + i, "caption": c} for i, c in enumerate(
        ["A dog sits on the street.", "a Dog, and a cat!", "The cat sits.", "a rare zebra"] + ["a dog"] * 3)]
    senti = [{"filename": "COCO_val2014_000000000001.jpg", "sentences": [{"raw": "A lovely dog"}, {"raw": "lovely lovely cat"}]},
             {"filename": "COCO_val2014_000000000999.jpg", "sentences": [{"raw": "ignored words here"}]}]
    assert caption_words("A Dog, and a cat!") == ["a", "dog", "and", "a", "cat"]
    assert simple_tokenize("wait... (really)") == ["wait", "...", "(", "really", ")"]
    words = build_caption_vocabulary(coco, senti, word_count_threshold=2, senticap_word_count_threshold=2)
    # COCO counts: a 6, dog 5, sits 2, cat 2, the 2 kept; on/street/and/rare/zebra 1 dropped; "lovely" 3x in SentiCap -> added
    assert words == sorted(["a", "dog", "sits", "cat", "the", "lovely"])
    (tmp_path / "c.json").write_text(json.dumps({"annotations": coco}))
    (tmp_path / "s.json").write_text(json.dumps({"images": senti}))
    vocab = build_from_files(str(tmp_path / "c.json"), str(tmp_path / "s.json"), str(tmp_path / "v"), 2, 2)
    assert vocab[:2] == ["@@UNKNOWN@@", "@@BOUNDARY@@"] and vocab[2:] == words
    assert open(tmp_path / "v" / "non_padded_namespaces.txt").read() == "tokens"
    v = Vocabulary.from_files(str(tmp_path / "v"))
    assert v.get_vocab_size() == len(vocab) and v.get_token_index("dog") == vocab.index("dog") and v.get_token_index("zebra") == 0
