"""ORACLE (test infrastructure): co-noir-spartan's public lookup round restated with Python integers:

  hash_tuple                      co-noir-spartan/co-spartan/src/worker.rs:836-846
  boost_degree / map_poly         co-noir-spartan/spartan/src/utils.rs:11-27
  LogLookupProof::prove           co-noir-spartan/spartan/src/logup.rs:31-80 (phi = x + t, h_0 = m / phi_0, h_1 = 1 / phi_1,
                                  h_0 and phi_0 boosted to the query's dimension)
  append_sumcheck_polys           co-noir-spartan/co-spartan/src/sumcheck.rs:459-500 (six products per lookup)
  partial_generate_eq             co-spartan/src/utils.rs (eq(z, .) restricted to a chunk; ark's little-endian order)
  IPForMLSumcheck::{prove_round}  ark-linear-sumcheck 0.5 @ nulltea/sumcheck e0500b8 (out of tree; restated from upstream
                                  knowledge: evaluations at t = 0 .. max_multiplicands, fix_variables on index bit 0)
  distributed_sumcheck_worker     co-spartan/src/worker.rs:694-724 + obtain_distrbuted_sumcheck_prover_state
                                  (sumcheck.rs:434-452)"""
import pyref as O

R = O.R


def hash_tuple(v, eq, v_msg):
    res = [(i + v_msg * eq[i]) % R for i in v if i is not None]
    n = len(res)
    npow = 1 << max(0, (n - 1).bit_length())
    return res + [res[0]] * (npow - n)


def boost_degree(g, new_dim):
    nv = len(g).bit_length() - 1
    assert new_dim >= nv
    factor = pow(pow(2, new_dim - nv, R), -1, R)
    scaled = [x * factor % R for x in g]
    return scaled * (1 << (new_dim - nv))


def eq_le(z):
    """generate_eq (spartan/src/utils.rs:38-52): ark's little-endian order -- index bit i pairs with z[i]"""
    ev = [1]
    for zi in z:
        ev = [e * (1 - zi) % R for e in ev] + [e * zi % R for e in ev]
    return ev


def loglookup_prove(query, table, m, x):
    """LogLookupProof::prove without the commitments -> ([h_0, h_1], [phi_0, phi_1])"""
    qv = len(query).bit_length() - 1
    phi_0 = [(x + t) % R for t in table]
    phi_1 = [(x + q) % R for q in query]
    h_0 = [mv * pow(p, -1, R) % R for mv, p in zip(m, phi_0)]
    h_0 = boost_degree(h_0, qv)
    phi_0 = boost_degree(phi_0, qv)
    h_1 = [pow(p, -1, R) for p in phi_1]
    return [h_0, h_1], [phi_0, phi_1]


def append_sumcheck_polys(polys, products, h, phi, m, degree_diff, z, lam):
    """sumcheck.rs:459-500 with the whole domain in one chunk (start = 0, log_chunk_size = num_vars): appends the flattened
    polynomials to `polys` and (coef, [indices]) products to `products`"""
    lagrange = eq_le(z)

    def add(p):
        polys.append(p)
        return len(polys) - 1

    il, ih0, ip0, im, ih1, ip1 = add(lagrange), add(h[0]), add(phi[0]), add(m), add(h[1]), add(phi[1])
    eta = lam
    products.append((lam % R, [ih0]))
    eta = eta * lam % R
    products.append((eta, [il, ih0, ip0]))
    products.append(((-eta) * pow(pow(2, degree_diff, R), -1, R) % R, [il, im]))
    products.append(((-lam) % R, [ih1]))
    eta = eta * lam % R
    products.append((eta, [il, ih1, ip1]))
    products.append(((-eta) % R, [il]))


def prove_round(polys, products, degree):
    """IPForMLSumcheck::prove_round's sums for the current tables"""
    half = len(polys[0]) // 2
    sums = [0] * (degree + 1)
    for b in range(half):
        for coef, idxs in products:
            prod = [coef] * (degree + 1)
            for j in idxs:
                start = polys[j][2 * b]
                step = (polys[j][2 * b + 1] - start) % R
                for t in range(degree + 1):
                    prod[t] = prod[t] * start % R
                    start = (start + step) % R
            for t in range(degree + 1):
                sums[t] = (sums[t] + prod[t]) % R
    return sums


def fix_variables(polys, r):
    return [[(p[2 * i] + r * (p[2 * i + 1] - p[2 * i])) % R for i in range(len(p) // 2)] for p in polys]


def interpolate_uni(evals, x):
    """the verifier's interpolation of a round message given at 0 .. d (ark-linear-sumcheck interpolate_uni_poly)"""
    n = len(evals)
    acc = 0
    for i in range(n):
        num, den = 1, 1
        for j in range(n):
            if j != i:
                num = num * (x - j) % R
                den = den * (i - j) % R
        acc = (acc + evals[i] * num % R * pow(den, -1, R)) % R
    return acc


def distributed_sumcheck(polys, products, transcript):
    """distributed_sumcheck_worker + a single-worker coordinator: returns (messages, point, final poly values)"""
    degree = max(len(f) for _, f in products)
    nv = len(polys[0]).bit_length() - 1
    polys = [list(p) for p in polys]
    msgs, point = [], []
    for _ in range(nv):
        ev = prove_round(polys, products, degree)
        transcript.append_scalars(ev)
        r = transcript.challenge_scalar()
        msgs.append(ev)
        point.append(r)
        polys = fix_variables(polys, r)
    return msgs, point, [p[0] for p in polys]


def distributed_sumcheck_split(polys, products, transcript, log_workers):
    """distributed_sumcheck_worker x 2^k + distributed_sumcheck_coordinator (co-spartan/src/coordinator.rs:748-811): worker j holds
    chunk j (the high k variables = j) of every polynomial, the coordinator sums the workers' messages of the first nv - k rounds and
    proves the last k rounds itself on the workers' final states (obtain_distrbuted_sumcheck_prover_state, sumcheck.rs:434-452).
    Returns (messages, point, final poly values) -- equal to distributed_sumcheck's by linearity of the round sums."""
    degree = max(len(f) for _, f in products)
    nv = len(polys[0]).bit_length() - 1
    K = 1 << log_workers
    cn = len(polys[0]) // K
    assert cn >= 2
    chunks = [[list(p[j * cn:(j + 1) * cn]) for p in polys] for j in range(K)]
    msgs, point = [], []
    for _ in range(nv - log_workers):
        ev = [0] * (degree + 1)
        for j in range(K):
            part = prove_round(chunks[j], products, degree)
            ev = [(a + b) % R for a, b in zip(ev, part)]
        transcript.append_scalars(ev)
        r = transcript.challenge_scalar()
        msgs.append(ev)
        point.append(r)
        chunks = [fix_variables(c, r) for c in chunks]
    merged = [[chunks[j][i][0] for j in range(K)] for i in range(len(polys))]  # polynomial i over the worker index
    for _ in range(log_workers):
        ev = prove_round(merged, products, degree)
        transcript.append_scalars(ev)
        r = transcript.challenge_scalar()
        msgs.append(ev)
        point.append(r)
        merged = fix_variables(merged, r)
    return msgs, point, [p[0] for p in merged]


def verify_sumcheck(msgs, point, finals, products, claimed_sum):
    expected = claimed_sum % R
    for ev, r in zip(msgs, point):
        if (ev[0] + ev[1]) % R != expected:
            return False
        expected = interpolate_uni(ev, r)
    total = 0
    for coef, idxs in products:
        v = coef
        for j in idxs:
            v = v * finals[j] % R
        total = (total + v) % R
    return total == expected
