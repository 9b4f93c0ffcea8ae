"""The step either side of the forward path: A3M alignment -> (msa, seq, aa_idx) tensors, and
(logits, xyz, plddt) -> distance / orientation maps and a PDB backbone.

The reference leaves both to the user (its README.md:20-51 feeds random integers in 0..20 and stops at the raw
outputs), so there is no reference code to restate here.  Conventions, stated so a caller can check them:

  * tokens 0..20 = "ARNDCQEGHILKMFPSTWYV-" (the 20 amino acids in the order of the official RoseTTAFold, gap / unknown = 20);
    `d_input=21` of the reference's embeddings (rf.py:106-120) is this alphabet size.
  * A3M: lower-case letters are insertions relative to the query and are dropped; '.' is treated as '-'; every kept
    row must then have the query's length.  Row 0 is the query, as `MsaEmbedding` assumes (rf.py:115-119).
  * aa_idx: residue numbers 0..L-1, plus `chain_break_offset` (200) after every chain break, so that the sequence-
    separation features (rf.py:177-181, 746-749) see the chains as far apart.
  * output bins (trRosetta / official RoseTTAFold 6D convention; last bin = "no contact"): dist 36 x 0.5 A over 2..20 A
    (+1); omega, theta 36 x 10 deg over -180..180 (+1); phi 18 x 10 deg over 0..180 (+1) -- matching the reference's head
    widths 37 / 37 / 37 / 19 (rf.py:1130-1172).

Everything here is host-side plumbing around the HIP path (plain torch / Python, any device); nothing in it is timed.
"""
import math

import torch

ALPHABET = "ARNDCQEGHILKMFPSTWYV-"
GAP = 20
_TOK = {c: i for i, c in enumerate(ALPHABET)}
_THREE = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER",
          "THR", "TRP", "TYR", "VAL", "UNK"]

DIST_MIN, DIST_MAX, DIST_BINS = 2.0, 20.0, 36
ANGLE_BINS, PHI_BINS = 36, 18


def tokenize(sequence):
    """Upper-case one-letter codes -> int64 tokens; anything outside the 20 amino acids (X, B, Z, '-', '.') -> 20."""
    return torch.tensor([_TOK.get(c, GAP) for c in sequence.upper()], dtype=torch.long)


def parse_a3m(text, max_seqs=None, dedup=True):
    """A3M text (or an open file / iterable of lines) -> (msa [N, L] int64, names list).  Row 0 is the query.
    Raises ValueError on an empty alignment or a row whose match-state length differs from the query's."""
    if hasattr(text, "read"):
        text = text.read()
    lines = text.splitlines() if isinstance(text, str) else list(text)
    names, seqs, cur = [], [], None
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.startswith("#"):
            continue
        if ln.startswith(">"):
            names.append(ln[1:].strip())
            seqs.append([])
            cur = seqs[-1]
        else:
            if cur is None:  # bare sequence without a header
                names.append("")
                seqs.append([])
                cur = seqs[-1]
            cur.append(ln)
    rows, kept_names, seen = [], [], set()
    for name, parts in zip(names, seqs):
        s = "".join(parts)
        s = "".join(c for c in s if not c.islower()).replace(".", "-")  # drop insertions
        if not s:
            continue
        if rows and len(s) != len(rows[0]):
            raise ValueError(f"parse_a3m: row '{name}' has {len(s)} match states, the query has {len(rows[0])}")
        if dedup and s in seen:
            continue
        seen.add(s)
        rows.append(s)
        kept_names.append(name)
        if max_seqs is not None and len(rows) >= max_seqs:
            break
    if not rows:
        raise ValueError("parse_a3m: no sequences")
    return torch.stack([tokenize(s) for s in rows]), kept_names


def residue_index(length, chain_lengths=None, chain_break_offset=200):
    """aa_idx [L]: 0..L-1 with `chain_break_offset` added after each chain boundary (chain_lengths sums to L)."""
    idx = torch.arange(length, dtype=torch.long)
    if chain_lengths:
        if sum(chain_lengths) != length:
            raise ValueError(f"residue_index: chain lengths {chain_lengths} do not sum to {length}")
        start = 0
        for k, cl in enumerate(chain_lengths):
            idx[start:start + cl] += k * chain_break_offset
            start += cl
    return idx


def featurize(a3m, n_seq=None, chain_lengths=None, chain_break_offset=200, device=None, max_len=None):
    """A3M text -> the three inputs of `RoseTTAFold.forward` (rf.py:1273) with a batch axis of 1:
    msa [1, N, L], seq [1, L] (= msa row 0), aa_idx [1, L].  n_seq keeps the first n_seq distinct rows.
    max_len: the model's positional-encoding table size (rf.py:57-76); a larger residue index raises here instead of in
    the kernel's range check."""
    msa, _ = parse_a3m(a3m, max_seqs=n_seq)
    L = msa.shape[1]
    aa_idx = residue_index(L, chain_lengths, chain_break_offset)
    if max_len is not None and int(aa_idx.max()) >= max_len:
        raise IndexError(f"featurize: residue index {int(aa_idx.max())} does not fit the model's max_len={max_len}")
    out = (msa[None], msa[:1].clone(), aa_idx[None])
    if device is not None:
        out = tuple(t.to(device) for t in out)
    return out


def collate(samples, pad_token=GAP):
    """List of (msa [1,N_i,L], seq [1,L], aa_idx [1,L]) of EQUAL L -> one batch; shallower MSAs are padded with all-gap
    rows (the reference has no masking: every row it is given takes part in the attention, so padding rows are a
    modelling choice of the caller -- this helper makes it explicit)."""
    L = samples[0][0].shape[-1]
    if any(s[0].shape[-1] != L for s in samples):
        raise ValueError("collate: samples must share the sequence length (the reference has no length masking)")
    n = max(s[0].shape[1] for s in samples)
    msas = []
    for m, _, _ in samples:
        pad = torch.full((1, n - m.shape[1], L), pad_token, dtype=m.dtype, device=m.device)
        msas.append(torch.cat([m, pad], 1))
    return torch.cat(msas), torch.cat([s[1] for s in samples]), torch.cat([s[2] for s in samples])


# ------------------------------------------------------------------------------------------------ decoding
def dist_bin_centers(device=None):
    step = (DIST_MAX - DIST_MIN) / DIST_BINS
    return DIST_MIN + step * (torch.arange(DIST_BINS, device=device, dtype=torch.float32) + 0.5)


def angle_bin_centers(n_bins, lo, device=None):
    step = 2 * math.pi / ANGLE_BINS
    return lo + step * (torch.arange(n_bins, device=device, dtype=torch.float32) + 0.5)


def decode_logits(logits, contact_cutoff=8.0):
    """logits dict of the forward (theta/phi/dist/omega [B,L,L,bins]) -> dict:
        p_contact   [B,L,L]  P(d < contact_cutoff)
        p_no_contact[B,L,L]  probability of the last ("> 20 A") distance bin
        dist_argmax [B,L,L]  most likely distance bin (int64; 36 = no contact)
        dist_expected [B,L,L]  E[d | contact] over the 36 distance bins (A)
        omega, theta, phi [B,L,L]  circular-mean angle (radians) over the non-"no contact" bins."""
    out = {}
    pd = torch.softmax(logits["dist"].float(), -1)
    centers = dist_bin_centers(pd.device)
    upper = centers + 0.5 * (DIST_MAX - DIST_MIN) / DIST_BINS
    out["p_contact"] = pd[..., :DIST_BINS][..., upper <= contact_cutoff + 1e-6].sum(-1)
    out["p_no_contact"] = pd[..., DIST_BINS]
    out["dist_argmax"] = pd.argmax(-1)
    within = pd[..., :DIST_BINS]
    out["dist_expected"] = (within * centers).sum(-1) / within.sum(-1).clamp_min(1e-8)
    for name, nb, lo in (("omega", ANGLE_BINS, -math.pi), ("theta", ANGLE_BINS, -math.pi), ("phi", PHI_BINS, 0.0)):
        pa = torch.softmax(logits[name].float(), -1)[..., :nb]
        c = angle_bin_centers(nb, lo, pa.device)
        if name == "phi":  # polar angle in [0, pi]: plain expectation
            out[name] = (pa * c).sum(-1) / pa.sum(-1).clamp_min(1e-8)
        else:  # dihedrals: circular mean
            out[name] = torch.atan2((pa * torch.sin(c)).sum(-1), (pa * torch.cos(c)).sum(-1))
    return out


def to_pdb(xyz, seq, plddt=None, aa_idx=None, chain_lengths=None):
    """Backbone (N, CA, C) of ONE sample as PDB text.  xyz [L,3,3] (atom order N, CA, C as in rf.py:1287-1289),
    seq [L] tokens, plddt [L] in 0..1 (written as the B-factor x 100), aa_idx [L] residue numbers (default 1..L)."""
    xyz = torch.as_tensor(xyz).detach().float().cpu()
    seq = torch.as_tensor(seq).detach().cpu().tolist()
    L = xyz.shape[0]
    if xyz.shape != (L, 3, 3) or len(seq) != L:
        raise ValueError(f"to_pdb: xyz {tuple(xyz.shape)} / seq {len(seq)} do not describe one L-residue backbone")
    b = [0.0] * L if plddt is None else (torch.as_tensor(plddt).detach().float().cpu().clamp(0, 1) * 100).tolist()
    resnum = list(range(1, L + 1)) if aa_idx is None else [int(v) + 1 for v in torch.as_tensor(aa_idx).cpu().tolist()]
    chain_of = ["A"] * L
    if chain_lengths:
        start = 0
        for k, cl in enumerate(chain_lengths):
            for i in range(start, start + cl):
                chain_of[i] = chr(ord("A") + k % 26)
            start += cl
    lines, serial = [], 1
    for i in range(L):
        name3 = _THREE[seq[i]] if 0 <= seq[i] < 20 else "UNK"
        for a, (atom, elem) in enumerate((("N", "N"), ("CA", "C"), ("C", "C"))):
            x, y, z = xyz[i, a].tolist()
            lines.append("ATOM  %5d  %-3s %3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s" %
                         (serial, atom, name3, chain_of[i], resnum[i] % 10000, x, y, z, 1.0, b[i], elem))
            serial += 1
        if chain_lengths and i + 1 < L and chain_of[i + 1] != chain_of[i]:
            lines.append("TER")
    lines += ["TER", "END"]
    return "\n".join(lines) + "\n"


def from_pdb_backbone(text):
    """Inverse of to_pdb for tests / round trips: PDB text -> (xyz [L,3,3], seq [L] tokens, bfactor [L])."""
    order = {"N": 0, "CA": 1, "C": 2}
    three = {n: i for i, n in enumerate(_THREE)}
    res, cur_key = [], None
    for ln in text.splitlines():
        if not ln.startswith("ATOM"):
            continue
        atom = ln[12:16].strip()
        if atom not in order:
            continue
        key = (ln[21], ln[22:26])
        if key != cur_key:
            res.append({"name": ln[17:20], "b": float(ln[60:66]), "xyz": [[float("nan")] * 3 for _ in range(3)]})
            cur_key = key
        res[-1]["xyz"][order[atom]] = [float(ln[30:38]), float(ln[38:46]), float(ln[46:54])]
    xyz = torch.tensor([r["xyz"] for r in res], dtype=torch.float32)
    seq = torch.tensor([min(three.get(r["name"], 20), 20) for r in res], dtype=torch.long)
    return xyz, seq, torch.tensor([r["b"] for r in res], dtype=torch.float32)
