"""Caption vocabulary files - counterpart of var_updown/scripts/build_vocabulary.py:55-136.

Output format (read back by ssc_runtime.vocab.Vocabulary.from_files and by allennlp's Vocabulary.from_files in the
reference): `tokens.txt` with "@@UNKNOWN@@" (id 0 = padding id) and "@@BOUNDARY@@" (id 1) first, then the kept words in
sorted order, one per line; `non_padded_namespaces.txt` containing "tokens".

Word selection (build_vocabulary.py:55-107): a COCO caption word is kept when it occurs at least `word_count_threshold`
times; a word of a SentiCap sentence (only sentences of images that also occur in the COCO list) is added when its
SentiCap count plus its COCO count reaches `senticap_word_count_threshold`.

Tokenisation: the reference lower-cases, strips, runs nltk's `word_tokenize` and drops the punctuation tokens listed in
PUNCTUATIONS.  nltk is not installable here, so the tokeniser is a parameter: `simple_tokenize` (default) splits on
whitespace after detaching the same punctuation marks - identical to the reference's result on plain COCO-style captions,
not on text where the Treebank tokeniser would split clitics ("don't" -> "do", "n't"); pass `nltk.tokenize.word_tokenize`
where nltk exists to reproduce the reference's files exactly."""
import json
import os
import re
from typing import Callable, Dict, Iterable, List, Optional

PUNCTUATIONS: List[str] = ["''", "'", "``", "`", "(", ")", "{", "}", ".", "?", "!", ",", ":", "-", "--", "...", ";"]
SPECIAL_TOKENS: List[str] = ["@@UNKNOWN@@", "@@BOUNDARY@@"]

_PUNCT_RE = re.compile(r"(\.\.\.|--|''|``|[`'(){}.?!,:;-])")


def simple_tokenize(text: str) -> List[str]:
    return _PUNCT_RE.sub(r" \1 ", text).split()


def caption_words(caption: str, tokenize: Callable[[str], List[str]] = simple_tokenize) -> List[str]:
    """Lower-cased, stripped, tokenised, punctuation tokens removed (build_vocabulary.py:69-72; the same normalisation the
    caption readers apply before the vocabulary lookup)."""
    return [t for t in tokenize(caption.lower().strip()) if t not in PUNCTUATIONS]


def build_caption_vocabulary(caption_json: Iterable[Dict], senticap_json: Optional[Iterable[Dict]] = None,
                             word_count_threshold: int = 5, senticap_word_count_threshold: int = 2,
                             tokenize: Callable[[str], List[str]] = simple_tokenize) -> List[str]:
    counts: Dict[str, int] = {}
    ids = set()
    for item in caption_json:
        ids.add(item["id"])   # (the reference collects the ANNOTATION id here and later compares it with a COCO image id)
        for w in caption_words(item["caption"], tokenize):
            counts[w] = counts.get(w, 0) + 1
    senti: Dict[str, int] = {}
    for item in senticap_json or ():
        coco_id = int(item["filename"].split(".")[0].split("_")[2])
        if coco_id not in ids:
            continue
        for sent in item["sentences"]:
            for w in caption_words(sent["raw"], tokenize):
                senti[w] = senti.get(w, 0) + 1
    kept = {w for w, c in counts.items() if c >= word_count_threshold}
    for w, c in senti.items():
        if c + counts.get(w, 0) >= senticap_word_count_threshold:
            kept.add(w)
    return sorted(kept)


def write_vocabulary(words: List[str], output_dirpath: str) -> List[str]:
    vocab = SPECIAL_TOKENS + list(words)
    os.makedirs(output_dirpath, exist_ok=True)
    with open(os.path.join(output_dirpath, "tokens.txt"), "w") as f:
        for w in vocab:
            f.write(w + "\n")
    with open(os.path.join(output_dirpath, "non_padded_namespaces.txt"), "w") as f:
        f.write("tokens")
    return vocab


def build_from_files(captions_jsonpath: str, senticap_jsonpath: Optional[str], output_dirpath: str, word_count_threshold: int = 5,
                     senticap_word_count_threshold: int = 2, tokenize: Callable[[str], List[str]] = simple_tokenize) -> List[str]:
    captions = json.load(open(captions_jsonpath))["annotations"]
    senticap = json.load(open(senticap_jsonpath))["images"] if senticap_jsonpath else None
    return write_vocabulary(build_caption_vocabulary(captions, senticap, word_count_threshold, senticap_word_count_threshold,
                                                     tokenize), output_dirpath)
