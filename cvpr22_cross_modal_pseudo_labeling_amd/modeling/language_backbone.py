"""Text side of the cross-modal head: the reference's ``BERT`` module
(maskrcnn_benchmark/modeling/language_backbone/transformers.py:7-79) as the student-teacher detector uses it
(modeling/detector/st_generalized_rcnn.py:45, 190-191, 202-209, 242).

What the reference computes is NOT a BERT forward pass: the strings are WordPiece-tokenised and the rows of BERT's input
word-embedding table (``bert_model.embeddings.word_embeddings.weight``, kept as the parameter ``embeddings`` -- state-dict
key ``bert.embeddings``) are looked up; ``extract_emb`` then averages the rows of a string's real tokens ([CLS], [SEP]
and padding excluded through ``special_tokens_mask``) and L2-normalises.  Here

* the table is the same parameter under the same name, so a reference checkpoint's ``bert.embeddings`` loads into it
  (its ``bert.bert_model.*`` transformer weights are never used by this path and have no counterpart);
* the tokenizer is HuggingFace's ``BertTokenizer`` (the reference's dependency, transformers==3.0.2 there) built from a
  local WordPiece vocabulary file -- ``vocab_file=``, ``$OVIS_BERT_VOCAB`` or the local HF cache of
  ``bert-base-uncased`` -- at first use; nothing is downloaded;
* ``extract_emb`` is ONE kernel launch (``_C.text_embed``: gather + masked mean + normalise, no [N, L, D] tensor) and is
  cached per (strings, table version): the reference re-extracts the 1203 LVIS names every iteration
  (st_generalized_rcnn.py:190-191) although the table is frozen.
"""
import os
import threading

import torch
from torch import nn


def normalize_class_names(names):
    """data/datasets/helper/parser.py:10-20: '_', '/', '(' and ')' become spaces, lower case."""
    out = []
    for name in names:
        for ch in "_/()":
            name = name.replace(ch, " ")
        out.append(name.lower())
    return out


# The extract_emb cache is read and re-ordered by two threads: PipelinedTrainer's worker (forward_frozen -> noun embeddings)
# and the training thread (forward_student -> prepare_text).  One module-level lock (an instance attribute would make the
# module un-deepcopy-able; the critical sections are a few dictionary operations).
_CACHE_LOCK = threading.Lock()


class BERT(nn.Module):
    VOCAB_SIZE, HIDDEN_SIZE = 30522, 768  # bert-base-uncased (BertConfig.from_pretrained, transformers.py:11)
    CACHE_ENTRIES = 64  # extract_emb results kept: the vocabulary names + the recent per-image noun lists

    def __init__(self, cfg=None, vocab_file=None, vocab_size=None, hidden_size=None, tokenizer=None):
        super().__init__()
        lb = getattr(getattr(cfg, "MODEL", None), "LANGUAGE_BACKBONE", None)
        ft_emb = bool(lb.FT_EMB) if lb is not None else False  # transformers.py:24: requires_grad = FT_EMB
        v, h = vocab_size or self.VOCAB_SIZE, hidden_size or self.HIDDEN_SIZE
        self.embeddings = nn.Parameter(torch.empty(v, h).normal_(0.0, 0.02), requires_grad=ft_emb)
        self.out_channels = h
        self.mlm = False  # transformers.py:34 asserts it
        self._vocab_file = vocab_file
        self._tokenizer = tokenizer
        self._cache = {}

    # -- tokenizer ------------------------------------------------------------------------------------------------
    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from transformers import BertTokenizer
            path = self._vocab_file or os.environ.get("OVIS_BERT_VOCAB")
            if path:
                self._tokenizer = BertTokenizer(path, do_lower_case=True)
            else:
                tok, err = None, None
                try:
                    tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
                except Exception as e:  # no cache, no network
                    err = e
                # (recent transformers hand back a tokenizer holding only the five special tokens when nothing is cached)
                if tok is None or len(tok) < 1000:
                    raise RuntimeError("BERT: no WordPiece vocabulary -- pass vocab_file=, set OVIS_BERT_VOCAB to a vocab.txt "
                                       "or provide a local HuggingFace cache of bert-base-uncased") from err
                self._tokenizer = tok
            if len(self._tokenizer) > self.embeddings.shape[0]:
                raise RuntimeError(f"BERT: the vocabulary has {len(self._tokenizer)} entries, the embedding table "
                                   f"{self.embeddings.shape[0]} rows")
        return self._tokenizer

    def tokenize(self, text_list):
        """``tokenizer.batch_encode_plus(text_list, add_special_tokens=True, pad_to_max_length=True,
        return_special_tokens_mask=True)`` (transformers.py:28-32) as int64 CPU tensors: [CLS] w1 .. wn [SEP] [PAD]*,
        special_tokens_mask 1 on [CLS] / [SEP] / [PAD]."""
        enc = self.tokenizer(list(text_list), add_special_tokens=True, padding=True, return_special_tokens_mask=True)
        return {k: torch.tensor(v, dtype=torch.int64) for k, v in enc.items()}

    def forward(self, text_list):
        """The reference's return value (transformers.py:59-68): the tokenizer's fields on the table's device plus
        ``input_embeddings`` [N, L, D] = ``embeddings[input_ids]``."""
        out = {k: v.to(self.embeddings.device) for k, v in self.tokenize(text_list).items()}
        out["input_embeddings"] = self.embeddings[out["input_ids"]]
        return out

    def extract_emb(self, words):
        """st_generalized_rcnn.py:202-209: [len(words), D] unit-norm embeddings (mean over the real tokens of every
        string, F.normalize) -- one launch, cached per (strings, table version)."""
        from .. import _C
        words = tuple(words)
        table = self.embeddings
        if table.requires_grad and torch.is_grad_enabled():
            # MODEL.LANGUAGE_BACKBONE.FT_EMB: the table is being trained, so the embeddings must stay in the graph -- the
            # reference's own tensor-op formula (st_generalized_rcnn.py:202-209), uncached (the table moves every step)
            enc = self.tokenize(words)
            ids = enc["input_ids"].to(table.device)
            keep = (1 - enc["special_tokens_mask"]).to(table.device, torch.float32)
            emb = (table[ids] * keep[:, :, None]).sum(1) / keep.sum(1)[:, None]
            return torch.nn.functional.normalize(emb, dim=-1)
        enc = None
        if table.requires_grad:
            # FT_EMB outside a graph (the @no_grad frozen half, eval): the table moves every optimizer step, and the fused
            # optimizer writes it through raw pointers, so ``_version`` does not see the update -- never serve a cached value
            enc = self.tokenize(words)
            return _C.text_embed(table.detach(), enc["input_ids"], enc["special_tokens_mask"])
        key = (words, table._version, table.device, table.data_ptr())
        with _CACHE_LOCK:
            hit = self._cache.get(key)
            if hit is not None:
                self._cache[key] = self._cache.pop(key)  # most recently used last
                return hit
        enc = self.tokenize(words)  # (host table, MODEL.DEVICE cpu: _C.text_embed dispatches to the in-package host formula)
        emb = _C.text_embed(table.detach(), enc["input_ids"], enc["special_tokens_mask"])
        # keyed by the strings: the per-image noun lists of a step must not evict the 1203-name vocabulary entry (the
        # reference re-tokenises it every iteration, st_generalized_rcnn.py:190-191).  A new table version drops everything.
        with _CACHE_LOCK:
            for k in [k for k in self._cache if k[1:] != key[1:]]:
                del self._cache[k]
            while len(self._cache) >= self.CACHE_ENTRIES:
                # evict the least recently used entry, sparing the LONGEST list: that one is the caption vocabulary
                longest = max(self._cache, key=lambda k: len(k[0]))
                victim = next((k for k in self._cache if k is not longest), longest)
                del self._cache[victim]
            self._cache[key] = emb
        return emb
