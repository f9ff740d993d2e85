"""Pure-Python restatement of the reference's unigram tokenizer (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/tokenizer.cpp:49-127 statement by statement (trie walk, utf-8 step logic,
unknown handling, backwards walk with the result-reference quirk), small inputs only.  tokenizer.cpp
cannot be compiled here (it includes ggml headers through util.h); the restatement is pinned instead to
the implementation the reference's converter reads the vocabulary FROM — Hugging Face `tokenizers`:
`models.Unigram` behind T5's pre-tokenizer for UnigramOracle (tests/golden/upstream_unigram.npz: 43 sentences
with doubled spaces and characters outside the vocabulary, identical ids), `models.BPE` with the byte-level
pre-tokenizer for BpeOracle (upstream_bpe.npz; a doubled space is the one stated divergence) —
tests/test_upstream_golden.py — and is what pins the C++ host tokenizers (tests/test_host_cpu.py).
"""
import math
import re

_DUPED_SPACES = re.compile(r"\s{2,}", re.ASCII)  # static std::regex duped_spaces("\\s{2,}") (tokenizer.h:22)
_UTF8_LEN = [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4]


class UnigramOracle:
    def __init__(self, vocab, scores, unk_token, eos_token=1):
        self.scores = [float(s) for s in scores]
        self.unk, self.eos = unk_token, eos_token
        self.unk_score = self.scores[unk_token]
        m = {}
        for i, t in enumerate(vocab):  # unordered_map<string,uint32_t>: later duplicates overwrite
            m[t.encode("utf-8", "surrogateescape") if isinstance(t, str) else bytes(t)] = i
        self.trie = {}
        for gram, tok in m.items():  # token_trie::add (tokenizer.cpp:3-22)
            node = self.trie
            for ch in gram:
                node = node.setdefault(ch, {})
            node["tok"] = tok

    def tokenize(self, text):
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        norm = b" " + _DUPED_SPACES.sub(" ", raw.decode("latin-1")).encode("latin-1")  # byte-wise like std::regex on char
        n = len(norm)
        res = [[self.unk, 0, -math.inf] for _ in range(n + 1)]
        res[0] = [self.unk, 0, 0.0]
        off = 0
        f32 = _f32
        while off < n:
            cur = off
            step = min(_UTF8_LEN[norm[off] >> 4], n - off)
            found_unknown = True
            best = res[off]
            node = self.trie.get(norm[cur])
            cur += 1
            while cur <= n and node is not None:
                if "tok" in node:
                    if cur - off == step:
                        found_unknown = False
                    score = f32(best[2] + self.scores[node["tok"]])
                    if score > res[cur][2]:
                        res[cur] = [node["tok"], off, score]
                node = node.get(norm[cur]) if cur < n else node.get(0)  # std::string[size()] == '\0'
                cur += 1
            if found_unknown:
                cur = off + step
                score = f32(best[2] + self.unk_score)
                if score > res[cur][2]:
                    res[cur] = [self.unk, off, score]
            off += step
        out = []
        prev_unknown = False
        r = res[n]
        while True:
            unknown = r[0] == self.unk
            if not (prev_unknown and unknown):
                out.append(r[0])
            if r[1] == 0:
                break
            prev_unknown = unknown
            r = res[r[1]]
        out.reverse()
        return out


def _f32(x):
    import struct
    return struct.unpack("f", struct.pack("f", x))[0] if math.isfinite(x) else x


class BpeOracle:
    """Pure-Python restatement of the reference's byte-pair tokenizer (src/tokenizer.cpp:209-296), written from its
    priority-queue formulation: pieces cut at spaces, "Ġ" prefix once a space has been seen, whole-piece lookup, then merges
    popped lowest (rank, left position) first and re-checked for staleness the way bpe_merge's size test does."""

    def __init__(self, vocab, merges):
        self.ids = {}
        for i, t in enumerate(vocab):
            self.ids[t] = i
        self.ranks = {}
        for i, m in enumerate(merges):
            a, b = m.split(" ")
            self.ranks[(a, b)] = i

    def tokenize(self, text):
        out = []
        space_prior = False
        for chunk in _split_keep(text, " "):
            if chunk != " ":
                self._piece(("Ġ" + chunk) if space_prior else chunk, out)
            else:
                space_prior = True
        return out

    def _piece(self, chunk, out):
        import heapq
        if chunk in self.ids:
            out.append(self.ids[chunk])
            return
        raw = chunk.encode("utf-8")
        parts = []  # [pos, size, prev, next]
        i = 0
        while i < len(raw):
            n = 1
            while i + n < len(raw) and (raw[i + n] & 0xC0) == 0x80:
                n += 1
            parts.append([i, n, len(parts) - 1, -1])
            i += n
        for k in range(len(parts) - 1):
            parts[k][3] = k + 1

        def s(k):
            return raw[parts[k][0]:parts[k][0] + parts[k][1]].decode("utf-8")

        heap = []

        def push(a, b):
            r = self.ranks.get((s(a), s(b)))
            if r is not None:
                heapq.heappush(heap, (r, parts[a][0], a, b, parts[a][1] + parts[b][1]))

        for k in range(len(parts) - 1):   # add_merges(only_forward = true)
            push(k, k + 1)
        while heap:
            r, _, a, b, new_size = heapq.heappop(heap)
            if parts[a][1] > 0 and parts[b][1] > 0 and new_size == parts[a][1] + parts[b][1] and parts[a][3] == b:
                parts[a][1] += parts[b][1]
                parts[b][1] = -1
                parts[a][3] = parts[b][3]
                if parts[a][3] >= 0:
                    parts[parts[a][3]][2] = a
                if parts[a][2] >= 0:
                    push(parts[a][2], a)
                if parts[a][3] >= 0:
                    push(a, parts[a][3])
        k = 0
        while k >= 0:
            out.append(self.ids.get(s(k), 0))
            k = parts[k][3]


def _split_keep(text, sep):
    """split(text, " ", true) of the reference's util: pieces and the separators between them, empty pieces dropped"""
    out, cur = [], ""
    for ch in text:
        if ch == sep:
            if cur:
                out.append(cur)
                cur = ""
            out.append(sep)
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out


# ---- Kokoro: single-pass tokenizer and clause chunking (TEST INFRASTRUCTURE) -------------------------------------------
class SinglePassOracle:
    """single_pass_tokenizer::tokenize (/root/reference/src/tokenizer.cpp:159-177): at each position, prefixes of 1, 2, ... bytes
    (up to the longest vocabulary entry) are looked up in order and the first hit wins; no hit -> id 0, one byte skipped."""

    def __init__(self, tokens):
        self.tokens = [t.encode("utf-8") if isinstance(t, str) else t for t in tokens]
        self.max_size = max((len(t) for t in self.tokens), default=0)

    def tokenize(self, text, out=None):
        rem = text.encode("utf-8") if isinstance(text, str) else text
        ids = [] if out is None else out
        while rem:
            tid = 0
            for i in range(1, min(len(rem) + 1, self.max_size + 1)):
                part = rem[:i]
                if part in self.tokens:
                    tid = self.tokens.index(part)
                    rem = rem[i:]
                    break
            if tid == 0:
                rem = rem[1:]
            ids.append(tid)
        return ids


def kokoro_chunks(tok, phonemes, max_ctx, space_id=16, bos=0, eos=0):
    """kokoro_runner::generate's split (src/models/kokoro/model.cpp:1420-1446) with tokenize_chunks (:1340-1388).  The
    reference reads chunks.back() of an empty list when the first clause is too long; that case counts as zero here."""
    p = (phonemes.encode("utf-8") if isinstance(phonemes, str) else phonemes).replace(b"\n", b" ")
    if len(p) < max_ctx - 2:
        for ch in b".!?":
            p = p.replace(bytes([ch]), b"")
        p = p.strip(b" ")
        if not p:
            return []
        return [[bos] + tok.tokenize(p) + [eos]]
    clauses, cur = [], b""
    for ch in p:
        if ch in b".!?":
            if cur:
                clauses.append(cur)
            cur = b""
        else:
            cur += bytes([ch])
    if cur:
        clauses.append(cur)
    chunks = []
    for clause in clauses:
        clause = clause.strip(b" ")
        if not clause:
            continue
        tokens = [bos] + tok.tokenize(clause)
        if len(tokens) > max_ctx - 2:
            last_space, last_split = 1, 1
            for i in range(1, len(tokens)):
                if tokens[i] == space_id:
                    last_space = i
                prev = len(chunks[-1]) if chunks else 0
                if (i - last_split) + prev >= max_ctx - 1:
                    if last_space > last_split:
                        chunks.append([bos] + tokens[last_split:last_space] + [eos])
                        last_split = last_space
                    else:
                        chunks.append([bos] + tokens[last_split:i + 1] + [eos])
                        last_split = i + 1
            if last_split + 1 < len(tokens):
                chunks.append([bos] + tokens[last_split:] + [eos])
        else:
            chunks.append(tokens + [eos])
    return chunks
