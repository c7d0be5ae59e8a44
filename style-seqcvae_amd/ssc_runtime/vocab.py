"""Minimal Vocabulary with the duck-type the hot path uses from allennlp.data.Vocabulary
(get_vocab_size / get_token_index / get_token_from_index / get_token_to_index_vocabulary) and the on-disk format
written by the reference's var_updown/scripts/build_vocabulary.py:122-136: `tokens.txt` (one token per line,
"@@UNKNOWN@@" first, "@@BOUNDARY@@" second) + `non_padded_namespaces.txt` containing "tokens"."""
import os
from typing import Dict, Iterable, List

UNKNOWN = "@@UNKNOWN@@"
BOUNDARY = "@@BOUNDARY@@"


class Vocabulary:
    def __init__(self, tokens: Iterable[str]):
        self._tokens: List[str] = list(tokens)
        if len(self._tokens) < 2 or self._tokens[0] != UNKNOWN or self._tokens[1] != BOUNDARY:
            raise ValueError(f"vocabulary must start with {UNKNOWN}, {BOUNDARY}")
        self._index: Dict[str, int] = {}
        for i, t in enumerate(self._tokens):
            self._index.setdefault(t, i)

    @classmethod
    def from_files(cls, directory: str) -> "Vocabulary":
        with open(os.path.join(directory, "tokens.txt"), encoding="utf-8") as f:
            tokens = [line.rstrip("\n") for line in f if line.rstrip("\n") != ""]
        return cls(tokens)

    @classmethod
    def synthetic(cls, size: int) -> "Vocabulary":
        return cls([UNKNOWN, BOUNDARY] + [f"w{i}" for i in range(2, size)])

    def save_to_files(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "tokens.txt"), "w", encoding="utf-8") as f:
            for t in self._tokens:
                f.write(t + "\n")
        with open(os.path.join(directory, "non_padded_namespaces.txt"), "w") as f:
            f.write("tokens")

    def add_token_to_namespace(self, token: str, namespace: str = "tokens") -> int:
        if token not in self._index:
            self._index[token] = len(self._tokens)
            self._tokens.append(token)
        return self._index[token]

    def get_vocab_size(self, namespace: str = "tokens") -> int:
        return len(self._tokens)

    def get_token_index(self, token: str, namespace: str = "tokens") -> int:
        return self._index.get(token, self._index[UNKNOWN])

    def get_token_from_index(self, index: int, namespace: str = "tokens") -> str:
        return self._tokens[index]

    def get_token_to_index_vocabulary(self, namespace: str = "tokens") -> Dict[str, int]:
        return dict(self._index)

    def get_index_to_token_vocabulary(self, namespace: str = "tokens") -> Dict[int, str]:
        return dict(enumerate(self._tokens))
