"""ORACLE (test infrastructure -- never imported by the product): exact big-integer restatement of the toggled / sparse
batched grand product of co-jolt's instruction lookups, following the reference line by line:

  Rep3SparseInterleavedPolynomial          co-jolt/src/poly/sparse_interleaved_poly.rs:28-196 (new, coalesce, uninterleave,
                                           layer_output), :198-380 (bind), :415-720 (compute_cubic, final_claims)
  Rep3BatchedGrandProductToggleLayer       co-jolt/src/subprotocols/sparse_grand_product.rs:30-135 (new, layer_output,
                                           coalesce), :137-290 (bind), :292-830 (compute_cubic: the four cases, final_claims)
  Rep3ToggledBatchedGrandProduct           sparse_grand_product.rs:832-1020 (construct, claimed_outputs, layers, prove_layer of
                                           the toggle layer) with the generic drivers of grand_product.rs:56-217 and
                                           sumcheck.rs:96-165

Shares are pyref's: a Rep3 share is a tuple (a, b), a plain value an int ("plain prover": one party, party index 0).
The SPARSE algorithms are restated as they are -- (index, value) lists, missing = a share of one, deltas against the
all-ones sums -- so that the engine's dense-equivalent kernels are checked against the reference's own formulation.
Out of tree (jolt-core, restated from upstream knowledge, parity unpinned): SparseCoefficient, the plain verifier of the
toggled grand product (ToggledBatchedGrandProduct::verify_sumcheck_claim)."""
import pyref as O

R = O.R


# ------------------------------------------------------------------------------------------------ share helpers
def one_share(party, nparties):
    """rep3::arithmetic::promote_to_trivial_share(party_id, F::one()) (types.rs:90-96); the plain prover's one is 1"""
    return 1 if nparties == 1 else O.rep3_promote_from_trivial(1, party)


def zero_share(nparties):
    return 0 if nparties == 1 else (0, 0)


def rep3_add_public(x, c, party):
    """rep3::arithmetic::add_public: P0 adds to a, P1 adds to b"""
    if not isinstance(x, tuple):
        return (x + c) % R
    if party == 0:
        return ((x[0] + c) % R, x[1])
    if party == 1:
        return (x[0], (x[1] + c) % R)
    return x


def rep3_sub_shared_by_public(x, c, party):
    return rep3_add_public(x, (-c) % R, party)


def additive_add_public(x, c, party):
    """additive::add_public (mpc-core/src/protocols/additive.rs): P0 only"""
    return (x + c) % R if party == 0 else x % R


def additive_sub_shared_by_public(x, c, party):
    return (x - c) % R if party == 0 else x % R


def _eq3(e0, e1):
    m = (e1 - e0) % R
    e2 = (e1 + m) % R
    return (e0, e2, (e2 + m) % R)


def _log2(n):
    assert n > 0 and n & (n - 1) == 0
    return n.bit_length() - 1


# ------------------------------------------------------------------------------------------------ sparse layer
class SparseLayer:
    """Rep3SparseInterleavedPolynomial (sparse_interleaved_poly.rs:28-75) of ONE party"""

    def __init__(self, coeffs, dense_len, party, nparties):
        batch_size = len(coeffs)
        per = dense_len // batch_size
        assert per & (per - 1) == 0
        self.party, self.nparties = party, nparties
        self.one = one_share(party, nparties)
        self.dense_len = dense_len
        if per <= 2:  # coalesce (:57-67)
            co = [self.one] * dense_len
            for seg in coeffs:
                for idx, val in seg:
                    co[idx] = val
            self.coeffs = [[] for _ in range(batch_size)]
            self.coalesced = co
        else:
            self.coeffs = [list(seg) for seg in coeffs]
            self.coalesced = None

    def batch_size(self):
        return len(self.coeffs)

    def coalesce(self):
        """:91-103"""
        if self.coalesced is not None:
            return list(self.coalesced)
        co = [self.one] * self.dense_len
        for seg in self.coeffs:
            for idx, val in seg:
                co[idx] = val
        return co

    def uninterleave(self):
        """:107-124"""
        if self.coalesced is not None:
            return O.interleaved_uninterleave(self.coalesced)
        left = [self.one] * (self.dense_len // 2)
        right = [self.one] * (self.dense_len // 2)
        for seg in self.coeffs:
            for idx, val in seg:
                if idx % 2 == 0:
                    left[idx // 2] = val
                else:
                    right[idx // 2] = val
        return left, right

    def layer_output_plan(self):
        """the sparse half of layer_output (:148-192): per segment a list of ('mul', right, left, index) /
        ('ready', index, value) -- FutureVal::pending_mul_args / Ready"""
        plan = []
        for seg in self.coeffs:
            out = []
            nxt = 0
            for j, (idx, val) in enumerate(seg):
                if idx < nxt:
                    continue
                if idx % 2 == 0:
                    right = seg[j + 1] if j + 1 < len(seg) else (idx + 1, self.one)
                    if right[0] == idx + 1:
                        out.append(("mul", right[1], val, idx // 2))
                    else:
                        out.append(("ready", idx // 2, val))
                    nxt = idx + 2
                else:
                    out.append(("ready", idx // 2, val))
                    nxt = idx + 1
            plan.append(out)
        return plan

    def bind(self, r):
        """Rep3Bindable::bind (:210-380)"""
        party = self.party
        if self.coalesced is not None:
            padded_len = (self.dense_len + 3) // 4 * 4
            self.coalesced = O.interleaved_bind(self.coalesced, r)
            self.dense_len = padded_len // 2
            return
        one = self.one

        def lerp(a, b):  # add_mul_public(a, b - a, r)
            return O.sh_add(a, O.sh_mul_public(O.sh_sub(b, a), r))

        def one_plus_r_times_minus_one(v):  # add_public(mul_public(sub_shared_by_public(v, 1), r), 1)
            return rep3_add_public(O.sh_mul_public(rep3_sub_shared_by_public(v, 1, party), r), 1, party)

        for s, seg in enumerate(self.coeffs):
            seg = list(seg)
            nl = nr = 0
            bound = 0
            for j in range(len(seg)):
                cidx, cval = seg[j]
                if cidx % 2 == 0 and cidx < nl:
                    continue
                if cidx % 2 == 1 and cidx < nr:
                    continue
                neighbors = [seg[j + 1] if j + 1 < len(seg) else (cidx + 1, one),
                             seg[j + 2] if j + 2 < len(seg) else (cidx + 2, one)]

                def find(q):
                    for ni, nv in neighbors:
                        if ni == q:
                            return nv
                    return one

                m = cidx % 4
                if m == 0:
                    seg[bound] = (cidx // 2, lerp(cval, find(cidx + 2)))
                    nl = cidx + 4
                elif m == 1:
                    if nl <= cidx + 1:
                        ln = seg[j + 1][1] if (j + 1 < len(seg) and seg[j + 1][0] == cidx + 1) else None
                        if ln is not None:
                            seg[bound] = (cidx // 2, one_plus_r_times_minus_one(ln))
                            bound += 1
                        nl = cidx + 3
                    seg[bound] = (cidx // 2 + 1, lerp(cval, find(cidx + 2)))
                    nr = cidx + 4
                elif m == 2:
                    seg[bound] = (cidx // 2 - 1, one_plus_r_times_minus_one(cval))
                    nl = cidx + 2
                else:
                    seg[bound] = (cidx // 2, one_plus_r_times_minus_one(cval))
                    nr = cidx + 2
                bound += 1
            self.coeffs[s] = seg[:bound]
        self.dense_len //= 2
        if self.dense_len // self.batch_size() == 2:
            self.coalesced = self.coalesce()

    def compute_cubic_evals(self, eq, previous_claim):
        """compute_cubic (:415-715) -> [g(0), claim - g(0), g(2), g(3)] additive"""
        party = self.party
        if self.coalesced is not None:
            return O.interleaved_compute_cubic_evals(self.coalesced, eq, previous_claim)
        one = self.one

        def block_terms(block):
            left = (block[0], block[2])
            right = (block[1], block[3])
            ml, mr = O.sh_sub(left[1], left[0]), O.sh_sub(right[1], right[0])
            l2 = O.sh_add(left[1], ml)
            l3 = O.sh_add(l2, ml)
            r2 = O.sh_add(right[1], mr)
            r3 = O.sh_add(r2, mr)
            return O.sh_local_mul(left[0], right[0]), O.sh_local_mul(l2, r2), O.sh_local_mul(l3, r3)

        def blocks_of(seg):  # chunk_by(index / 4)
            out, cur = [], []
            for c in seg:
                if cur and cur[0][0] // 4 != c[0] // 4:
                    out.append(cur)
                    cur = []
                cur.append(c)
            if cur:
                out.append(cur)
            return out

        if eq.E1_len == 1:
            eq_evals = [_eq3(eq.E2[2 * k], eq.E2[2 * k + 1]) for k in range(min(len(eq.E2) // 2, self.dense_len // 4))]
            sums = [sum(e[i] for e in eq_evals) % R for i in range(3)]
            deltas = [0, 0, 0]
            for seg in self.coeffs:
                for sb in blocks_of(seg):
                    bi = sb[0][0] // 4
                    block = [one] * 4
                    for idx, val in sb:
                        block[idx % 4] = val
                    t = block_terms(block)
                    ee = eq_evals[bi]
                    for i in range(3):
                        deltas[i] = (deltas[i] + additive_sub_shared_by_public(t[i] * ee[i] % R, ee[i], party)) % R
            ev = [additive_add_public(deltas[i], sums[i], party) for i in range(3)]
        else:
            E1e = [_eq3(eq.E1[2 * j], eq.E1[2 * j + 1]) for j in range(eq.E1_len // 2)]
            E1s = [sum(e[i] for e in E1e) % R for i in range(3)]
            nbits = _log2(eq.E1_len) - 1
            mask = (1 << nbits) - 1
            deltas = [0, 0, 0]
            for seg in self.coeffs:
                # group by x2
                groups, cur = [], []
                for c in seg:
                    if cur and (cur[0][0] // 4) >> nbits != (c[0] // 4) >> nbits:
                        groups.append(cur)
                        cur = []
                    cur.append(c)
                if cur:
                    groups.append(cur)
                for g in groups:
                    inner = [0, 0, 0]
                    for sb in blocks_of(g):
                        bi = sb[0][0] // 4
                        block = [one] * 4
                        for idx, val in sb:
                            block[idx % 4] = val
                        t = block_terms(block)
                        x1 = bi & mask
                        for i in range(3):
                            inner[i] = (inner[i] + additive_sub_shared_by_public(t[i], 1, party) * E1e[x1][i]) % R
                    x2 = (g[0][0] // 4) >> nbits
                    for i in range(3):
                        deltas[i] = (deltas[i] + inner[i] * eq.E2[x2]) % R
            dl = self.dense_len
            if dl & (dl - 1) == 0:
                ones = list(E1s)
            else:
                npow = 1 << (dl - 1).bit_length()
                chunk_size = npow // eq.E2_len
                nfull = dl // chunk_size
                e2sum = sum(eq.E2[:nfull]) % R
                if dl % chunk_size == 0:
                    ones = [e2sum * E1s[i] % R for i in range(3)]
                else:
                    last = E1e[:(dl % chunk_size) // 4]
                    ls = [sum(e[i] for e in last) % R for i in range(3)]
                    ones = [(e2sum * E1s[i] + eq.E2[nfull] * ls[i]) % R for i in range(3)]
            ev = [additive_add_public(deltas[i], ones[i], party) for i in range(3)]
        return [ev[0], (previous_claim - ev[0]) % R, ev[1], ev[2]]

    def final_claims(self):
        assert self.dense_len == 2
        d = self.coalesce()
        return d[0], d[1]


def sparse_layer_output(layers, mask_fn=None):
    """layer_output (:135-196) for all parties at once: the pending multiplications go through
    rep3::arithmetic::mul (local product + zero-sharing mask, then the ring reshare: c.b = previous party's c.a);
    mask_fn(n) -> three zero-sum mask vectors (None: zero masks -- they cancel in every opened value)"""
    nparties = len(layers)
    l0 = layers[0]
    if l0.coalesced is not None:
        if nparties == 1:
            nxt = [O.interleaved_layer_output_local(l0.coalesced)]
        else:
            lr = [O.interleaved_uninterleave(layers[p].coalesced) for p in range(3)]
            n = len(lr[0][0])
            masks = mask_fn(n) if mask_fn else [[0] * n] * 3
            nxt = O.rep3_mul_vec([x[0] for x in lr], [x[1] for x in lr], masks)
        out = []
        for p in range(nparties):
            o = SparseLayer([[] for _ in range(l0.batch_size())], 2 * l0.batch_size(), p, nparties)  # placeholder shape
            o.dense_len = l0.dense_len // 2
            o.coeffs = [[] for _ in range(l0.batch_size())]
            o.coalesced = nxt[p]
            out.append(o)
        return out
    plans = [layers[p].layer_output_plan() for p in range(nparties)]
    # the sparsity pattern is public: every party builds the same plan shape
    muls = [(s, k) for s, seg in enumerate(plans[0]) for k, e in enumerate(seg) if e[0] == "mul"]
    n = len(muls)
    masks = mask_fn(n) if (mask_fn and nparties == 3) else [[0] * n] * 3
    ca = [[(O.sh_local_mul(plans[p][s][k][1], plans[p][s][k][2]) + (masks[p][t] if nparties == 3 else 0)) % R
           for t, (s, k) in enumerate(muls)] for p in range(nparties)]
    out = []
    for p in range(nparties):
        segs = []
        t_of = {sk: t for t, sk in enumerate(muls)}
        for s, seg in enumerate(plans[p]):
            o = []
            for k, e in enumerate(seg):
                if e[0] == "mul":
                    t = t_of[(s, k)]
                    val = ca[p][t] if nparties == 1 else (ca[p][t], ca[(p + 2) % 3][t])
                    o.append((e[3], val))
                else:
                    o.append((e[1], e[2]))
            segs.append(o)
        out.append(SparseLayer(segs, l0.dense_len // 2, p, nparties))
    return out


# ------------------------------------------------------------------------------------------------ toggle layer
class ToggleLayer:
    """Rep3BatchedGrandProductToggleLayer (sparse_grand_product.rs:30-70) of ONE party.
    flag_indices: per PAIR of circuits the sorted indices of the set flags; fingerprints: per circuit N shares"""

    def __init__(self, flag_indices, fingerprints, party, nparties):
        self.party, self.nparties = party, nparties
        self.flag_indices = [list(f) for f in flag_indices]
        self.flag_values = []
        self.fingerprints = [list(f) for f in fingerprints]
        self.layer_len = 2 * len(fingerprints[0])
        self.batched_layer_len = len(fingerprints) * self.layer_len
        self.coalesced_flags = None
        self.coalesced_fingerprints = None

    def layer_output(self):
        """:76-97"""
        vals = []
        for b, fps in enumerate(self.fingerprints):
            vals.append([(b * self.layer_len // 2 + i, fps[i]) for i in self.flag_indices[b // 2]])
        return SparseLayer(vals, self.batched_layer_len // 2, self.party, self.nparties)

    def coalesce(self):
        """:104-134"""
        fps = [f[0] for f in self.fingerprints]
        npow = 1 << max(0, (len(fps) - 1).bit_length())
        fps += [zero_share(self.nparties)] * (npow - len(fps))
        flags = []
        for idxs, vals in zip(self.flag_indices, self.flag_values):
            assert len(idxs) <= 1
            co = [0, 0]
            for i, v in zip(idxs, vals):
                assert i == 0
                co = [v, v]
            flags += co
        npf = 1 << max(0, (len(flags) - 1).bit_length())
        flags += [1] * (npf - len(flags))
        self.coalesced_fingerprints, self.coalesced_flags = fps, flags

    def bind(self, r):
        """:153-290"""
        if self.coalesced_flags is not None:
            f = self.coalesced_flags
            self.coalesced_flags = [(f[2 * i] + r * (f[2 * i + 1] - f[2 * i])) % R for i in range(len(f) // 2)]
            p = self.coalesced_fingerprints
            self.coalesced_fingerprints = [O.sh_add(p[2 * i], O.sh_mul_public(O.sh_sub(p[2 * i + 1], p[2 * i]), r)) for i in range(len(p) // 2)]
            self.batched_layer_len //= 2
            return
        assert self.layer_len % 4 == 0
        n = self.layer_len // 4
        for layer in self.fingerprints:
            for i in range(n):
                layer[i] = O.sh_add(layer[2 * i], O.sh_mul_public(O.sh_sub(layer[2 * i + 1], layer[2 * i]), r))
        first = not self.flag_values
        if first:
            self.flag_values = [[] for _ in self.flag_indices]
        for idxs, vals in zip(self.flag_indices, self.flag_values):
            nxt = 0
            bound = 0
            for j in range(len(idxs)):
                index = idxs[j]
                if index < nxt:
                    continue
                idxs[bound] = index // 2
                if index % 2 == 0:
                    neighbor = idxs[j + 1] if j + 1 < len(idxs) else 0
                    if neighbor == index + 1:
                        if first:
                            vals.append(1)
                        else:
                            vals[bound] = (vals[j] + r * (vals[j + 1] - vals[j])) % R
                    else:
                        if first:
                            vals.append((1 - r) % R)
                        else:
                            vals[bound] = (vals[j] - r * vals[j]) % R
                    nxt = index + 2
                else:
                    if first:
                        vals.append(r % R)
                    else:
                        vals[bound] = r * vals[j] % R
                    nxt = index + 1
                bound += 1
            del idxs[bound:]
        self.layer_len //= 2
        self.batched_layer_len //= 2
        if self.layer_len == 2:
            self.coalesce()

    def _pairs(self, b):
        """the (flags, fingerprints) pairs of circuit b at the set flags, as cases 3 / 4 walk them (:487-545)"""
        fps = self.fingerprints[b]
        idxs = self.flag_indices[b // 2]
        unbound = not self.flag_values
        vals = None if unbound else self.flag_values[b // 2]
        nxt = 0
        for j, index in enumerate(idxs):
            if index < nxt:
                continue
            if index % 2 == 0:
                neighbor = idxs[j + 1] if j + 1 < len(idxs) else 0
                if neighbor == index + 1:
                    flags = (1, 1) if unbound else (vals[j], vals[j + 1])
                else:
                    flags = (1, 0) if unbound else (vals[j], 0)
                f = (fps[index], fps[index + 1])
                nxt = index + 2
            else:
                flags = (0, 1) if unbound else (0, vals[j])
                f = (fps[index - 1], fps[index])
                nxt = index + 1
            yield index, flags, f

    def compute_cubic_evals(self, eq, previous_claim):
        """compute_cubic (:311-823): the four cases"""
        party = self.party

        def terms(flags, fps):
            mf = (flags[1] - flags[0]) % R
            mp = O.sh_sub(fps[1], fps[0])
            f2 = (flags[1] + mf) % R
            f3 = (f2 + mf) % R
            p2 = O.sh_add(fps[1], mp)
            p3 = O.sh_add(p2, mp)
            return (flags[0], f2, f3), (O.sh_into_additive(fps[0]), O.sh_into_additive(p2), O.sh_into_additive(p3))

        if self.coalesced_flags is not None:
            cf, cp = self.coalesced_flags, self.coalesced_fingerprints
            s = [0, 0, 0]
            if eq.E1_len == 1:  # case 1
                n = min(len(cf) // 2, len(cp) // 2, len(eq.E2) // 2)
                for k in range(n):
                    ee = _eq3(eq.E2[2 * k], eq.E2[2 * k + 1])
                    fl, pa = terms((cf[2 * k], cf[2 * k + 1]), (cp[2 * k], cp[2 * k + 1]))
                    for i in range(3):
                        s[i] = (s[i] + additive_add_public(pa[i] * fl[i] % R, (1 - fl[i]) % R, party) * ee[i]) % R
            else:  # case 2
                E1e = [_eq3(eq.E1[2 * j], eq.E1[2 * j + 1]) for j in range(eq.E1_len // 2)]
                fcs = (1 << max(0, (len(cf) - 1).bit_length())) // eq.E2_len
                pcs = (1 << max(0, (len(cp) - 1).bit_length())) // eq.E2_len
                for x2 in range(eq.E2_len):
                    fx, px = cf[x2 * fcs:(x2 + 1) * fcs], cp[x2 * pcs:(x2 + 1) * pcs]
                    if not fx or not px:
                        break
                    inner = [0, 0, 0]
                    for j in range(min(len(E1e), len(fx) // 2, len(px) // 2)):
                        fl, pa = terms((fx[2 * j], fx[2 * j + 1]), (px[2 * j], px[2 * j + 1]))
                        for i in range(3):
                            inner[i] = (inner[i] + additive_add_public(pa[i] * fl[i] % R, (1 - fl[i]) % R, party) * E1e[j][i]) % R
                    for i in range(3):
                        s[i] = (s[i] + inner[i] * eq.E2[x2]) % R
            return [s[0], (previous_claim - s[0]) % R, s[1], s[2]]
        if eq.E1_len == 1:  # case 3
            eq_evals = [_eq3(eq.E2[2 * k], eq.E2[2 * k + 1]) for k in range(min(eq.E2_len // 2, self.batched_layer_len // 4))]
            sums = [sum(e[i] for e in eq_evals) % R for i in range(3)]
            deltas = [0, 0, 0]
            for b in range(len(self.fingerprints)):
                for index, flags, fps in self._pairs(b):
                    fl, pa = terms(flags, fps)
                    ee = eq_evals[(self.layer_len * b) // 4 + index // 2]
                    for i in range(3):
                        deltas[i] = (deltas[i] + additive_sub_shared_by_public(pa[i] * fl[i] % R, fl[i], party) * ee[i]) % R
            ev = [additive_add_public(deltas[i], sums[i], party) for i in range(3)]
        else:  # case 4
            E1e = [_eq3(eq.E1[2 * j], eq.E1[2 * j + 1]) for j in range(eq.E1_len // 2)]
            E1s = [sum(e[i] for e in E1e) % R for i in range(3)]
            nbits = _log2(eq.E1_len) - 1
            mask = (1 << nbits) - 1
            deltas = [0, 0, 0]
            for b in range(len(self.fingerprints)):
                delta = [0, 0, 0]
                inner = [0, 0, 0]
                prev_x2 = 0
                for index, flags, fps in self._pairs(b):
                    fl, pa = terms(flags, fps)
                    bi = (self.layer_len * b) // 4 + index // 2
                    x2 = bi >> nbits
                    if x2 != prev_x2:
                        for i in range(3):
                            delta[i] = (delta[i] + inner[i] * eq.E2[prev_x2]) % R
                        inner = [0, 0, 0]
                        prev_x2 = x2
                    x1 = bi & mask
                    for i in range(3):
                        inner[i] = (inner[i] + additive_sub_shared_by_public(pa[i] * fl[i] % R, fl[i], party) * E1e[x1][i]) % R
                for i in range(3):
                    delta[i] = (delta[i] + inner[i] * eq.E2[prev_x2]) % R
                    deltas[i] = (deltas[i] + delta[i]) % R
            bl = self.batched_layer_len
            if bl & (bl - 1) == 0:
                ones = list(E1s)
            else:
                npow = 1 << (bl - 1).bit_length()
                chunk_size = npow // eq.E2_len
                nfull = bl // chunk_size
                e2sum = sum(eq.E2[:nfull]) % R
                if bl % chunk_size == 0:
                    ones = [e2sum * E1s[i] % R for i in range(3)]
                else:
                    last = E1e[:(bl % chunk_size) // 4]
                    ls = [sum(e[i] for e in last) % R for i in range(3)]
                    ones = [(e2sum * E1s[i] + eq.E2[nfull] * ls[i]) % R for i in range(3)]
            ev = [additive_add_public(deltas[i], ones[i], party) for i in range(3)]
        return [ev[0], (previous_claim - ev[0]) % R, ev[1], ev[2]]

    def final_claims(self):
        """:825-835: (promote_to_trivial_share(flags[0]), fingerprints[0])"""
        assert self.layer_len == 2
        f = self.coalesced_flags[0]
        fl = f if self.nparties == 1 else O.rep3_promote_from_trivial(f, self.party)
        return fl, self.coalesced_fingerprints[0]


# ------------------------------------------------------------------------------------------------ toggled grand product
def toggled_construct(flag_indices, fingerprints_per_party, mask_fn=None):
    """Rep3ToggledBatchedGrandProduct::construct (:905-930): toggle layer + tree_depth sparse layers"""
    nparties = len(fingerprints_per_party)
    toggles = [ToggleLayer(flag_indices, fingerprints_per_party[p], p, nparties) for p in range(nparties)]
    tree_depth = _log2(len(fingerprints_per_party[0][0]))
    sparse = [[t.layer_output() for t in toggles]]
    for _ in range(tree_depth - 1):
        sparse.append(sparse_layer_output(sparse[-1], mask_fn))
    return toggles, sparse


def toggled_claimed_outputs(sparse):
    """:936-945"""
    outs = []
    for layer in sparse[-1]:
        left, right = layer.uninterleave()
        outs.append([O.sh_local_mul(l, r) for l, r in zip(left, right)])
    return outs


def _prove_layer_sumcheck(layers, eqs, claims, transcript, nparties):
    """prove_sumcheck / coordinate_prove_arbitrary (sumcheck.rs:96-165) over one layer held by every party"""
    num_rounds = eqs[0].get_num_vars()
    r_sumcheck, round_polys = [], []
    prev = list(claims)
    for _ in range(num_rounds):
        msgs = [O.unipoly_from_evals(layers[p].compute_cubic_evals(eqs[p], prev[p])) for p in range(nparties)]
        poly = O.combine_additive(msgs)
        comp = O.unipoly_compress(poly)
        transcript.append_scalars(comp)
        r_j = transcript.challenge_scalar()
        r_sumcheck.append(r_j)
        nxt = O.unipoly_eval(poly, r_j)
        for p in range(nparties):
            layers[p].bind(r_j)
            eqs[p].bind(r_j)
        prev = [O.additive_promote_from_trivial(nxt, p) for p in range(nparties)]
        round_polys.append(comp)
    finals = [layers[p].final_claims() for p in range(nparties)]
    if nparties == 3:
        left = sum(f[0][0] for f in finals) % R
        right = sum(f[1][0] for f in finals) % R
    else:
        left, right = finals[0][0] % R, finals[0][1] % R
    return r_sumcheck, round_polys, finals, left, right


def toggled_prove(toggles, sparse, transcript):
    """prove_grand_product_worker / cooridinate_prove_grand_product (grand_product.rs:56-130) over
    layers() = [toggle, sparse...].rev() (:947-960); the toggle layer's prove_layer has no r_layer fold (:850-873)"""
    nparties = len(toggles)
    outputs = O.combine_additive(toggled_claimed_outputs(sparse))
    transcript.append_scalars(outputs)
    padded = list(outputs)
    while len(padded) & (len(padded) - 1):
        padded.append(0)
    r = transcript.challenge_vector(len(padded).bit_length() - 1)
    claim_pub = sum(e * v for e, v in zip(O.eq_evals(r), padded)) % R
    claims = [O.additive_promote_from_trivial(claim_pub, p) for p in range(nparties)]
    proof = {"outputs": outputs, "layers": []}
    for layer in reversed(sparse):
        eqs = [O.SplitEq(r) for _ in range(nparties)]
        rs, polys, finals, left, right = _prove_layer_sumcheck(layer, eqs, claims, transcript, nparties)
        transcript.append_scalar(left)
        transcript.append_scalar(right)
        r = list(reversed(rs))
        r_layer = transcript.challenge_scalar()
        claims = [O.sh_into_additive(O.sh_add(finals[p][0], O.sh_mul_public(O.sh_sub(finals[p][1], finals[p][0]), r_layer)))
                  for p in range(nparties)]
        r.append(r_layer)
        proof["layers"].append({"round_polys": polys, "left": left, "right": right})
    eqs = [O.SplitEq(r) for _ in range(nparties)]
    rs, polys, finals, left, right = _prove_layer_sumcheck(toggles, eqs, claims, transcript, nparties)
    transcript.append_scalar(left)
    transcript.append_scalar(right)
    r = list(reversed(rs))
    proof["layers"].append({"round_polys": polys, "left": left, "right": right})
    return proof, r


def toggled_verify(proof, transcript):
    """plain verifier (jolt-core ToggledBatchedGrandProduct::verify_sumcheck_claim, out of tree): multiplication layers
    check eq * L * R and fold with r_layer; the toggle layer (last) checks eq * (flag * fingerprint + 1 - flag).
    Returns (flag claim, fingerprint claim, r) or None"""
    outputs = proof["outputs"]
    transcript.append_scalars(outputs)
    padded = list(outputs)
    while len(padded) & (len(padded) - 1):
        padded.append(0)
    r = transcript.challenge_vector(len(padded).bit_length() - 1)
    claim = sum(e * v for e, v in zip(O.eq_evals(r), padded)) % R
    nl = len(proof["layers"])
    for li, lp in enumerate(proof["layers"]):
        rs = []
        e = claim
        for comp in lp["round_polys"]:
            c1 = (e - 2 * comp[0] - sum(comp[1:])) % R
            poly = [comp[0], c1] + list(comp[1:])
            transcript.append_scalars(comp)
            r_j = transcript.challenge_scalar()
            rs.append(r_j)
            e = O.unipoly_eval(poly, r_j)
        if len(rs) != len(r):
            return None
        eqv = 1
        for a, b in zip(r, reversed(rs)):
            eqv = eqv * ((a * b + (1 - a) * (1 - b)) % R) % R
        left, right = lp["left"], lp["right"]
        transcript.append_scalar(left)
        transcript.append_scalar(right)
        r = list(reversed(rs))
        if li != nl - 1:
            if eqv * left % R * right % R != e:
                return None
            r_layer = transcript.challenge_scalar()
            claim = (left + r_layer * (right - left)) % R
            r.append(r_layer)
        else:
            if eqv * ((left * right + 1 - left) % R) % R != e:
                return None
    return proof["layers"][-1]["left"], proof["layers"][-1]["right"], r


def toggled_leaf_mles(flag_indices, fingerprints, r):
    """direct evaluation of the two leaf polynomials at the final point: the flags (each pair's flags serve both of its
    circuits) and the fingerprints, circuit-major, zero-padded to a power of two -- what the final claims must equal"""
    n = len(fingerprints[0])
    flags, fps = [], []
    for b, f in enumerate(fingerprints):
        s = set(flag_indices[b // 2])
        flags += [1 if i in s else 0 for i in range(n)]
        fps += list(f)
    npow = 1 << max(0, (len(fps) - 1).bit_length())
    flags += [1] * (npow - len(flags))  # flags are padded with ones, fingerprints with zeros (:118-131)
    fps += [0] * (npow - len(fps))
    eq = O.eq_evals(r)
    return (sum(e * v for e, v in zip(eq, flags)) % R, sum(e * v for e, v in zip(eq, fps)) % R)
