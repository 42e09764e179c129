"""ORACLE (test infrastructure -- never imported by the product): the Jolt RV32IM constraint SET of the reference as a
table of linear combinations, and a synthetic trace that satisfies it.

  JoltRV32IMConstraints::uniform_constraints / cross_step_constraints      co-jolt/src/r1cs/constraints.rs:39-257
  JoltR1CSInputs (variant order, flatten, get_ref: which column is shared)  co-jolt/src/r1cs/inputs.rs:189-328

Out of tree (jolt-core r1cs/builder.rs, r1cs/constraints.rs, jolt-common rv_trace.rs; restated from upstream knowledge,
parity unpinned -- the reference holds no serialized constraint system):
  R1CSBuilder::constrain_binary(v)                 v * (1 - v) = 0
  constrain_eq(l, r)                               (l - r) * 1 = 0
  constrain_eq_conditional(cond, l, r)             cond * (l - r) = 0
  constrain_if_else(cond, t, f, res)               cond * (t - f) = res - f        (allocate_if_else: res is an aux variable)
  constrain_prod(l, r, res)                        l * r = res                     (allocate_prod: res is an aux variable)
  pack_be(vars, bits)                              sum_i vars[i] * 2^(bits * (n - 1 - i))
  constants LOG_M = 16, OPERAND_SIZE = 8, PC_START_ADDRESS = 0x80000000, PC_NOOP_SHIFT = 4, REGISTER_COUNT = 64
  CircuitFlags: the 12 flags constraints.rs names; their enum ORDER is jolt-common's and cannot be read here -- a parameter.
`memory_start` is the memory layout's input_start (a per-program parameter).

An LC is a list of (variable index | None for the constant, i64 coefficient); a uniform constraint is (a, b, c) with
Az * Bz = Cz; a cross-step constraint is (a, b, cond) of (offset flag, LC) meaning cond * (a - b) = 0, offset = evaluated at
the next step (OffsetEqConstraint, spartan_interleaved_poly.rs:100-160)."""
import pyref as O

R = O.R

C = 4
LOG_M = 16
OPERAND_SIZE = 8
PC_START_ADDRESS = 0x80000000
PC_NOOP_SHIFT = 4
REGISTER_COUNT = 64
MEMORY_START = 0x7FFF8000

CIRCUIT_FLAGS = ["LeftOperandIsPC", "RightOperandIsImm", "Load", "Store", "Jump", "Branch", "WriteLookupOutputToRD", "Lui",
                 "ConcatLookupQueryChunks", "Virtual", "Assert", "DoNotUpdatePC"]
INSTRUCTIONS = ["ADD", "SUB", "AND", "OR", "XOR", "BEQ", "BGE", "BGEU", "BNE", "SLT", "SLTU", "SLL", "SRA", "SRL", "MOVSIGN", "MUL", "MULU",
                "MULHU", "VIRTUAL_ADVICE", "VIRTUAL_MOVE", "VIRTUAL_ASSERT_LTE", "VIRTUAL_ASSERT_VALID_SIGNED_REMAINDER",
                "VIRTUAL_ASSERT_VALID_UNSIGNED_REMAINDER", "VIRTUAL_ASSERT_VALID_DIV0", "VIRTUAL_ASSERT_HALFWORD_ALIGNMENT", "VIRTUAL_POW2",
                "VIRTUAL_SRA_PADDING"]  # jolt/vm/rv32i_vm.rs:41-70


def input_names():
    """JoltR1CSInputs::flatten::<4>() (inputs.rs:237-262) with AuxVariable::iter() in the order of AuxVariableStuff"""
    n = ["Bytecode_A", "Bytecode_ELFAddress", "Bytecode_Bitflags", "Bytecode_RS1", "Bytecode_RS2", "Bytecode_RD", "Bytecode_Imm", "RAM_Address",
         "RS1_Read", "RS2_Read", "RD_Read", "RAM_Read", "RD_Write", "RAM_Write"]
    n += ["ChunksQuery%d" % i for i in range(C)] + ["LookupOutput"]
    n += ["ChunksX%d" % i for i in range(C)] + ["ChunksY%d" % i for i in range(C)]
    n += ["Op_" + f for f in CIRCUIT_FLAGS] + ["I_" + f for f in INSTRUCTIONS]
    n += ["LeftLookupOperand", "RightLookupOperand", "Product"] + ["RelevantYChunk%d" % i for i in range(C)]
    n += ["WriteLookupOutputToRD", "WritePCtoRD", "NextPCJump", "ShouldBranch", "NextPC"]
    return n


NAMES = input_names()
IDX = {n: i for i, n in enumerate(NAMES)}
NUM_INPUTS = len(NAMES)  # 78

# which columns are public (inputs.rs:269-300 get_ref + the witness structs: bytecode a / v[0..5], a_ram, the flags and the
# two public aux products are public polynomials; bytecode v[5] = imm, the register / RAM values, dim, lookup outputs,
# operand chunks and the other aux variables are shared -- SURVEY App. A)
_PUBLIC = {"Bytecode_A", "Bytecode_ELFAddress", "Bytecode_Bitflags", "Bytecode_RS1", "Bytecode_RS2", "Bytecode_RD", "RAM_Address",
           "WriteLookupOutputToRD", "WritePCtoRD"} | {"Op_" + f for f in CIRCUIT_FLAGS} | {"I_" + f for f in INSTRUCTIONS}
IS_PUBLIC = [n in _PUBLIC for n in NAMES]
# compact storage of the public columns (what the commitment's small-scalar MSM consumes): bytes per entry
PUBLIC_BYTES = {"Bytecode_A": 4, "Bytecode_ELFAddress": 4, "Bytecode_Bitflags": 8, "Bytecode_RS1": 1, "Bytecode_RS2": 1, "Bytecode_RD": 1, "RAM_Address": 4,
                "WriteLookupOutputToRD": 1, "WritePCtoRD": 1}


# ------------------------------------------------------------------------------------------------ LC algebra
def lc(*terms):
    """terms: (name | None, coeff)"""
    return [(IDX[v] if v is not None else None, c) for v, c in terms if c != 0]


def lc_add(a, b):
    acc = {}
    order = []
    for v, c in a + b:
        if v not in acc:
            acc[v] = 0
            order.append(v)
        acc[v] += c
    return [(v, acc[v]) for v in order if acc[v] != 0]


def lc_scale(a, k):
    return [(v, c * k) for v, c in a]


def lc_sub(a, b):
    return lc_add(a, lc_scale(b, -1))


def pack_be(names, bits):
    n = len(names)
    return lc(*[(names[i], 1 << (bits * (n - 1 - i))) for i in range(n)])


def build_system():
    """-> (uniform, cross, padded rows per step).  Order = the order of the cs.* calls in constraints.rs:43-222"""
    U = []
    one = lc((None, 1))

    def binary(v):
        U.append((lc((v, 1)), lc_sub(one, lc((v, 1))), []))

    def eq_conditional(cond, left, right):
        U.append((cond, lc_sub(left, right), []))

    def if_else(cond, t, f, res):
        U.append((cond, lc_sub(t, f), lc_sub(lc((res, 1)), f)))
        return lc((res, 1))

    def prod(res, left, right):
        U.append((left, right, lc((res, 1))))
        return lc((res, 1))

    op = lambda f: lc(("Op_" + f, 1))
    ins = lambda f: lc(("I_" + f, 1))
    v = lambda n: lc((n, 1))

    for f in INSTRUCTIONS:
        binary("I_" + f)
    for f in CIRCUIT_FLAGS:
        binary("Op_" + f)
    flags = ["Op_" + f for f in CIRCUIT_FLAGS] + ["I_" + f for f in INSTRUCTIONS]
    U.append((lc_sub(pack_be(flags, 1), v("Bytecode_Bitflags")), one, []))  # constrain_pack_be = constrain_eq(packed, result)

    real_pc = lc(("Bytecode_ELFAddress", 4), (None, PC_START_ADDRESS - PC_NOOP_SHIFT))
    x = if_else(op("LeftOperandIsPC"), real_pc, v("RS1_Read"), "LeftLookupOperand")
    y = if_else(op("RightOperandIsImm"), v("Bytecode_Imm"), v("RS2_Read"), "RightLookupOperand")

    is_load_or_store = lc_add(op("Load"), op("Store"))
    eq_conditional(is_load_or_store, lc_add(v("RS1_Read"), v("Bytecode_Imm")),
                   lc(("RAM_Address", 4), (None, MEMORY_START - 4 * REGISTER_COUNT)))
    eq_conditional(op("Load"), v("RAM_Read"), v("RAM_Write"))
    eq_conditional(op("Load"), v("RAM_Read"), v("RD_Write"))
    eq_conditional(op("Store"), v("RS2_Read"), v("RAM_Write"))
    eq_conditional(op("Lui"), v("RD_Write"), v("Bytecode_Imm"))

    query = ["ChunksQuery%d" % i for i in range(C)]
    packed_query = pack_be(query, LOG_M)
    eq_conditional(lc_add(ins("ADD"), ins("VIRTUAL_ASSERT_HALFWORD_ALIGNMENT")), packed_query, lc_add(x, y))
    eq_conditional(ins("SUB"), packed_query, lc_add(lc_sub(x, y), lc((None, 0xFFFFFFFF + 1))))
    is_mul = lc_add(lc_add(ins("MUL"), ins("MULU")), ins("MULHU"))
    product = prod("Product", v("RS1_Read"), v("RS2_Read"))
    eq_conditional(is_mul, packed_query, product)
    eq_conditional(lc_add(ins("MOVSIGN"), ins("VIRTUAL_MOVE")), packed_query, x)
    eq_conditional(op("Assert"), v("LookupOutput"), one)

    xs = ["ChunksX%d" % i for i in range(C)]
    ys = ["ChunksY%d" % i for i in range(C)]
    eq_conditional(op("ConcatLookupQueryChunks"), pack_be(xs, OPERAND_SIZE), x)
    eq_conditional(op("ConcatLookupQueryChunks"), pack_be(ys, OPERAND_SIZE), y)
    is_shift = lc_add(lc_add(ins("SLL"), ins("SRL")), ins("SRA"))
    for i in range(C):
        rel = if_else(is_shift, v(ys[C - 1]), v(ys[i]), "RelevantYChunk%d" % i)
        eq_conditional(op("ConcatLookupQueryChunks"), v(query[i]), lc_add(lc((xs[i], 1 << 8)), rel))

    w1 = prod("WriteLookupOutputToRD", v("Bytecode_RD"), op("WriteLookupOutputToRD"))
    eq_conditional(w1, v("RD_Write"), v("LookupOutput"))
    w2 = prod("WritePCtoRD", v("Bytecode_RD"), op("Jump"))
    eq_conditional(w2, lc(("Bytecode_ELFAddress", 4), (None, PC_START_ADDRESS)), v("RD_Write"))

    next_pc_jump = if_else(op("Jump"), lc(("LookupOutput", 1), (None, 4)),
                           lc(("Bytecode_ELFAddress", 4), (None, PC_START_ADDRESS + 4), ("Op_DoNotUpdatePC", -4)), "NextPCJump")
    should_branch = prod("ShouldBranch", op("Branch"), v("LookupOutput"))
    if_else(should_branch, lc(("Bytecode_ELFAddress", 4), (None, PC_START_ADDRESS), ("Bytecode_Imm", 1)), next_pc_jump, "NextPC")

    # OffsetEqConstraint::new(cond, a, b) (constraints.rs:228-250) in the (a, b, cond) order of this oracle
    cross = [((False, v("NextPC")), (True, lc(("Bytecode_ELFAddress", 4), (None, PC_START_ADDRESS))), (True, v("Bytecode_ELFAddress"))),
             ((True, v("Bytecode_A")), (False, lc(("Bytecode_A", 1), (None, 1))), (False, op("Virtual")))]
    padded = 1
    while padded < len(U) + len(cross):
        padded <<= 1
    return U, cross, padded


# ------------------------------------------------------------------------------------------------ the synthetic trace
# step kinds (s_kind % 16); every path of the constraint set is taken by some kind
K_ALU, K_ALU_IMM, K_SHIFT, K_ADD, K_SUB, K_MUL, K_LOAD, K_STORE, K_LUI, K_JUMP, K_BRANCH, K_ASSERT, K_MOVE, K_ADDI, K_VIRT, K_ALIGN = range(16)
_ALU_INSTR = [2, 3, 4, 9, 10]      # AND OR XOR SLT SLTU
_SHIFT_INSTR = [11, 12, 13]        # SLL SRA SRL
_MUL_INSTR = [15, 16, 17]
_BRANCH_INSTR = [5, 6, 7, 8]
_ASSERT_INSTR = [20, 22, 23]


def n_padding(n):
    return max(1, n // 8)


def synthetic_columns(seed, n):
    """the dealer's view of the NUM_INPUTS columns (field elements) for a trace of n = 2^k steps; the last n_padding(n)
    steps are padding (all flags zero, ELF address 0).  Mirrored by csrc/host/jolt_r1cs.hpp jolt_build_clear."""
    sm = lambda off, bits: O.synthetic_small(seed + off, n, bits)
    s_kind, s_instr = sm(11, 8), sm(12, 8)
    s_rs1, s_rs2, s_imm, s_out = sm(13, 32), sm(14, 32), sm(15, 12), sm(16, 32)
    s_rd, s_regs, s_addr, s_misc, s_rdread, s_target = sm(17, 6), sm(18, 12), sm(19, 20), sm(20, 8), sm(21, 32), sm(22, 16)
    cols = [[0] * n for _ in range(NUM_INPUTS)]
    n_real = n - n_padding(n)

    def put(name, t, val):
        cols[IDX[name]][t] = val % R

    elf = 1 + (s_target[0] if n else 0)
    bca = 5
    for t in range(n):
        if t >= n_real:
            # padding: only the aux variables that the constraints derive from zeros are non-zero
            put("NextPCJump", t, PC_START_ADDRESS + 4)
            put("NextPC", t, PC_START_ADDRESS + 4)
            continue
        kind = s_kind[t] % 16
        if t == n_real - 1 and kind == K_VIRT:
            kind = K_ALU  # a virtual sequence never ends the trace (constraints.rs:240-245)
        F = {f: 0 for f in CIRCUIT_FLAGS}
        rs1, rs2, imm, out = s_rs1[t], s_rs2[t], s_imm[t], s_out[t]
        rd = s_rd[t]
        ram_addr = ram_read = ram_write = 0
        rd_write = s_rdread[t] ^ 0x5A5A5A5A
        cx = [0] * C
        cy = [0] * C
        q = [0] * C
        nxt_elf = None  # None: fall through to elf + 1 - DoNotUpdatePC
        sel = s_instr[t]
        if kind in (K_ALU, K_ALU_IMM, K_SHIFT):
            instr = _SHIFT_INSTR[sel % 3] if kind == K_SHIFT else _ALU_INSTR[sel % 5]
            F["ConcatLookupQueryChunks"] = 1
            F["WriteLookupOutputToRD"] = 1
            F["RightOperandIsImm"] = 1 if kind == K_ALU_IMM else 0
        elif kind in (K_ADD, K_ADDI):
            instr = 0
            F["WriteLookupOutputToRD"] = 1
            F["RightOperandIsImm"] = 1 if kind == K_ADDI else 0
        elif kind == K_SUB:
            instr = 1
            F["WriteLookupOutputToRD"] = 1
        elif kind == K_MUL:
            instr = _MUL_INSTR[sel % 3]
            F["WriteLookupOutputToRD"] = 1
        elif kind == K_LOAD:
            instr = 19  # the load's own lookup is a move of the loaded word
            F["Load"] = 1
            ram_addr = s_addr[t]
            rs1 = 4 * ram_addr + MEMORY_START - 4 * REGISTER_COUNT - imm
            ram_read = ram_write = rd_write = s_out[t]
        elif kind == K_STORE:
            instr = 19
            F["Store"] = 1
            ram_addr = s_addr[t]
            rs1 = 4 * ram_addr + MEMORY_START - 4 * REGISTER_COUNT - imm
            ram_read = s_rdread[t]
            ram_write = rs2
        elif kind == K_LUI:
            instr = 18
            F["Lui"] = 1
            F["RightOperandIsImm"] = 1
            rd_write = imm
        elif kind == K_JUMP:
            instr = 0
            F["Jump"] = 1
            F["LeftOperandIsPC"] = 1
            F["RightOperandIsImm"] = 1
            nxt_elf = 1 + s_target[t]
        elif kind == K_BRANCH:
            instr = _BRANCH_INSTR[sel % 4]
            F["Branch"] = 1
            F["ConcatLookupQueryChunks"] = 1
            out = s_misc[t] & 1
            if out:
                nxt_elf = 1 + s_target[t]
        elif kind == K_ASSERT:
            instr = _ASSERT_INSTR[sel % 3]
            F["Assert"] = 1
            F["ConcatLookupQueryChunks"] = 1
            out = 1
        elif kind == K_MOVE:
            instr = 14 if sel & 1 else 19
            F["WriteLookupOutputToRD"] = 1
        elif kind == K_VIRT:
            instr = 18
            F["Virtual"] = 1
            F["DoNotUpdatePC"] = 1
            F["WriteLookupOutputToRD"] = 1
        else:  # K_ALIGN
            instr = 24
            F["Assert"] = 1
            F["RightOperandIsImm"] = 1
            out = 1
        real_pc = 4 * elf + PC_START_ADDRESS - PC_NOOP_SHIFT
        if kind == K_JUMP:
            out = 4 * nxt_elf + PC_START_ADDRESS - 4  # the jump target, NextPCJump = LookupOutput + 4
            imm = out - real_pc                       # x + y = LookupOutput (a field element; imm is a shared column)
        if kind == K_BRANCH and nxt_elf is not None:
            imm = 4 * (nxt_elf - elf)
        x = real_pc if F["LeftOperandIsPC"] else rs1
        y = imm if F["RightOperandIsImm"] else rs2
        if F["ConcatLookupQueryChunks"]:
            # operands are 32-bit words here (x = rs1; y = rs2 or a 12-bit immediate)
            cx = [(x >> (8 * (C - 1 - i))) & 0xFF for i in range(C)]
            cy = [(y >> (8 * (C - 1 - i))) & 0xFF for i in range(C)]
            shift = instr in _SHIFT_INSTR
            rel = [cy[C - 1] if shift else cy[i] for i in range(C)]
            q = [cx[i] * 256 + rel[i] for i in range(C)]
        else:
            shift = False
            rel = [0] * C
            if instr in (0, 24):
                pq = (x + y) % R
            elif instr == 1:
                pq = (x - y + (1 << 32)) % R
            elif instr in _MUL_INSTR:
                pq = rs1 * rs2
            elif instr in (14, 19):
                pq = x % R
            else:
                pq = s_out[t] ^ 0x1234  # unconstrained by the R1CS
            assert 0 <= pq < (1 << 64)
            q = [(pq >> (LOG_M * (C - 1 - i))) & 0xFFFF for i in range(C)]
        if kind == K_JUMP and rd != 0:
            rd_write = 4 * elf + PC_START_ADDRESS
        if F["WriteLookupOutputToRD"] and rd != 0:
            rd_write = out
        # aux variables by their defining constraints
        w1 = rd * F["WriteLookupOutputToRD"]
        w2 = rd * F["Jump"]
        npj = out + 4 if F["Jump"] else 4 * elf + PC_START_ADDRESS + 4 - 4 * F["DoNotUpdatePC"]
        sb = F["Branch"] * out
        npc = (4 * elf + PC_START_ADDRESS + imm) if sb else npj
        if nxt_elf is None:
            nxt_elf = elf + 1 - F["DoNotUpdatePC"]
        assert (npc - PC_START_ADDRESS) % R == 4 * nxt_elf % R
        bits = [F[f] for f in CIRCUIT_FLAGS] + [1 if i == instr else 0 for i in range(len(INSTRUCTIONS))]
        bitflags = 0
        for b in bits:
            bitflags = 2 * bitflags + b
        put("Bytecode_A", t, bca)
        put("Bytecode_ELFAddress", t, elf)
        put("Bytecode_Bitflags", t, bitflags)
        put("Bytecode_RS1", t, s_regs[t] & 63)
        put("Bytecode_RS2", t, s_regs[t] >> 6)
        put("Bytecode_RD", t, rd)
        put("Bytecode_Imm", t, imm)
        put("RAM_Address", t, ram_addr)
        put("RS1_Read", t, rs1)
        put("RS2_Read", t, rs2)
        put("RD_Read", t, s_rdread[t])
        put("RAM_Read", t, ram_read)
        put("RD_Write", t, rd_write)
        put("RAM_Write", t, ram_write)
        for i in range(C):
            put("ChunksQuery%d" % i, t, q[i])
            put("ChunksX%d" % i, t, cx[i])
            put("ChunksY%d" % i, t, cy[i])
            put("RelevantYChunk%d" % i, t, rel[i])
        put("LookupOutput", t, out)
        for f in CIRCUIT_FLAGS:
            put("Op_" + f, t, F[f])
        put("I_" + INSTRUCTIONS[instr], t, 1)
        put("LeftLookupOperand", t, x)
        put("RightLookupOperand", t, y)
        put("Product", t, rs1 % R * (rs2 % R))
        put("WriteLookupOutputToRD", t, w1)
        put("WritePCtoRD", t, w2)
        put("NextPCJump", t, npj)
        put("ShouldBranch", t, sb)
        put("NextPC", t, npc)
        # the next step: virtual sequences continue at the next bytecode row, everything else lands anywhere
        bca = bca + 1 if F["Virtual"] else 7 + (s_misc[t] | (s_target[t] << 8))
        elf = nxt_elf
    return cols


def check_satisfied(uniform, cross, cols, n):
    """every row of Az * Bz - Cz is zero (the dealer's sanity check; also run by tests)"""
    import pyspartan_outer as S
    polys = [("P", c) for c in cols]
    for step in range(n):
        nxt = step + 1 if step + 1 < n else None
        for ci, (a, b, c) in enumerate(uniform):
            az = S.eval_lc(a, polys, step, 0)[1] if a else 0
            bz = S.eval_lc(b, polys, step, 0)[1] if b else 0
            cz = S.eval_lc(c, polys, step, 0)[1] if c else 0
            if (az * bz - cz) % R:
                return ("uniform", ci, step)
        for ci, (a, b, cond) in enumerate(cross):
            az = (S.eval_offset_lc(a, polys, step, nxt, 0)[1] - S.eval_offset_lc(b, polys, step, nxt, 0)[1]) % R
            bz = S.eval_offset_lc(cond, polys, step, nxt, 0)[1]
            if az * bz % R:
                return ("cross", ci, step)
    return None
