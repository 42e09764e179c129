// The Jolt RV32IM constraint SET of the reference as the harness's LC table, and a synthetic trace that satisfies it.
//
//   JoltRV32IMConstraints::uniform_constraints / cross_step_constraints     co-jolt/src/r1cs/constraints.rs:39-257
//   JoltR1CSInputs (variant order, flatten, get_ref: which column is shared)  co-jolt/src/r1cs/inputs.rs:189-328
//
// 70 uniform + 2 cross-step constraints over 78 inputs -> 128 rows per step, 128 padded variables.  The builder calls
// (constrain_binary / constrain_eq_conditional / allocate_if_else / allocate_prod / constrain_pack_be, pack_be) and the
// constants (LOG_M, OPERAND_SIZE, PC_START_ADDRESS, PC_NOOP_SHIFT, REGISTER_COUNT) live out of tree in jolt-core /
// jolt-common; they are restated from their published definitions (oracle/pyjolt_r1cs.py has the same table, the
// tests compare the two through whole proofs).  The ORDER of the 12 CircuitFlags inside jolt-common's enum cannot be read
// here: it only permutes columns.  The trace is synthetic (no Jolt tracer on the box): per step one of 16 instruction
// kinds, values chosen so that every constraint holds; the last max(1, N / 8) steps are padding.
#pragma once

namespace jolt {

constexpr int C = 4, LOG_M = 16, OPERAND_SIZE = 8, REGISTER_COUNT = 64, N_FLAGS = 12, N_INSTR = 27;
constexpr uint64_t PC_START_ADDRESS = 0x80000000ull, PC_NOOP_SHIFT = 4, MEMORY_START = 0x7FFF8000ull;

// JoltR1CSInputs::flatten::<4>() (inputs.rs:237-262)
enum Var : int {
    V_BYTECODE_A = 0, V_ELF, V_BITFLAGS, V_BC_RS1, V_BC_RS2, V_BC_RD, V_IMM, V_RAM_ADDR, V_RS1, V_RS2, V_RD_READ, V_RAM_READ, V_RD_WRITE, V_RAM_WRITE,
    V_QUERY = 14, V_OUT = 18, V_CX = 19, V_CY = 23, V_OP = 27, V_INSTR = 39, V_LEFT = 66, V_RIGHT, V_PRODUCT, V_REL = 69, V_W1 = 73, V_W2, V_NPJ, V_SB, V_NPC,
    NUM_INPUTS = 78
};
enum Flag : int { F_LEFT_PC = 0, F_RIGHT_IMM, F_LOAD, F_STORE, F_JUMP, F_BRANCH, F_WLO, F_LUI, F_CONCAT, F_VIRTUAL, F_ASSERT, F_DNU };
enum Instr : int { I_ADD = 0, I_SUB, I_AND, I_OR, I_XOR, I_BEQ, I_BGE, I_BGEU, I_BNE, I_SLT, I_SLTU, I_SLL, I_SRA, I_SRL, I_MOVSIGN, I_MUL, I_MULU, I_MULHU,
                   I_ADVICE, I_MOVE, I_ASSERT_LTE, I_ASSERT_SREM, I_ASSERT_UREM, I_ASSERT_DIV0, I_ASSERT_ALIGN, I_POW2, I_SRA_PAD };

// public columns and their compact width in bytes (0 = shared): bytecode a / v[0..4], a_ram, the flags, the two public aux
// products (inputs.rs:269-300 + the witness structs)
static inline int public_bytes(int v) {
    switch (v) {
        case V_BYTECODE_A: case V_ELF: case V_RAM_ADDR: return 4;
        case V_BITFLAGS: return 8;
        case V_BC_RS1: case V_BC_RS2: case V_BC_RD: case V_W1: case V_W2: return 1;
        default: return (v >= V_OP && v < V_LEFT) ? 1 : 0;
    }
}

typedef std::vector<std::pair<int, int64_t>> LC;  // (variable | -1 for the constant, coefficient)
static inline LC lc1(int v, int64_t c = 1) { return LC{{v, c}}; }
static inline LC operator+(LC a, const LC& b) {
    a.insert(a.end(), b.begin(), b.end());
    return a;
}
static inline LC operator-(LC a, const LC& b) {
    for (auto& t : b) a.push_back({t.first, -t.second});
    return a;
}
static inline LC pack_be(int first, int n, int bits, int stride = 1) {
    LC out;
    for (int i = 0; i < n; i++) out.push_back({first + i * stride, (int64_t)1 << (bits * (n - 1 - i))});
    return out;
}

struct System {
    std::vector<int> term_var;
    std::vector<int64_t> term_coeff;
    std::vector<cozk_lc> uniform, cross;
    // the same constraints as vectors (host: evaluate_matrix_mle_partial, the dealer's checks)
    struct Row {
        LC a, b, c;
        int off_a = 0, off_b = 0, off_c = 0;
    };
    std::vector<Row> u_rows, x_rows;
    cozk_r1cs desc{};
    size_t padded = 0, num_vars = 0;
    cozk_lc put(const LC& l, int offset = 0) {
        cozk_lc r{(int)term_var.size(), (int)l.size(), offset};
        for (auto& t : l) {
            term_var.push_back(t.first);
            term_coeff.push_back(t.second);
        }
        return r;
    }
    void add_uniform(const LC& a, const LC& b, const LC& c) {
        uniform.push_back(put(a));
        uniform.push_back(put(b));
        uniform.push_back(put(c));
        u_rows.push_back(Row{a, b, c});
    }
    // cond * (a - b) = 0; offsets: evaluated at the next step
    void add_cross(const LC& a, int off_a, const LC& b, int off_b, const LC& cond, int off_c) {
        cross.push_back(put(a, off_a));
        cross.push_back(put(b, off_b));
        cross.push_back(put(cond, off_c));
        x_rows.push_back(Row{a, b, cond, off_a, off_b, off_c});
    }
    void finish(size_t n_vars) {
        padded = 1;
        while (padded < u_rows.size() + x_rows.size()) padded <<= 1;
        num_vars = n_vars;
        desc.term_var = term_var.data();
        desc.term_coeff = term_coeff.data();
        desc.n_terms = term_var.size();
        desc.uniform = uniform.data();
        desc.n_uniform = u_rows.size();
        desc.cross = cross.data();
        desc.n_cross = x_rows.size();
        desc.padded_num_constraints = padded;
    }
};

// the order of the cs.* calls in constraints.rs:43-222, then cross_step_constraints (:225-256)
static inline void build_system(System& s) {
    const LC one = lc1(-1, 1), none;
    auto binary = [&](int v) { s.add_uniform(lc1(v), one - lc1(v), none); };
    auto eq_conditional = [&](const LC& cond, const LC& l, const LC& r) { s.add_uniform(cond, l - r, none); };
    auto if_else = [&](const LC& cond, const LC& t, const LC& f, int res) {
        s.add_uniform(cond, t - f, lc1(res) - f);
        return lc1(res);
    };
    auto prod = [&](int res, const LC& l, const LC& r) {
        s.add_uniform(l, r, lc1(res));
        return lc1(res);
    };
    auto op = [&](int f) { return lc1(V_OP + f); };
    auto ins = [&](int i) { return lc1(V_INSTR + i); };
    for (int i = 0; i < N_INSTR; i++) binary(V_INSTR + i);
    for (int f = 0; f < N_FLAGS; f++) binary(V_OP + f);
    s.add_uniform(pack_be(V_OP, N_FLAGS + N_INSTR, 1) - lc1(V_BITFLAGS), one, none);  // constrain_pack_be
    const LC real_pc = lc1(V_ELF, 4) + lc1(-1, (int64_t)(PC_START_ADDRESS - PC_NOOP_SHIFT));
    const LC x = if_else(op(F_LEFT_PC), real_pc, lc1(V_RS1), V_LEFT);
    const LC y = if_else(op(F_RIGHT_IMM), lc1(V_IMM), lc1(V_RS2), V_RIGHT);
    eq_conditional(op(F_LOAD) + op(F_STORE), lc1(V_RS1) + lc1(V_IMM), lc1(V_RAM_ADDR, 4) + lc1(-1, (int64_t)(MEMORY_START - 4 * REGISTER_COUNT)));
    eq_conditional(op(F_LOAD), lc1(V_RAM_READ), lc1(V_RAM_WRITE));
    eq_conditional(op(F_LOAD), lc1(V_RAM_READ), lc1(V_RD_WRITE));
    eq_conditional(op(F_STORE), lc1(V_RS2), lc1(V_RAM_WRITE));
    eq_conditional(op(F_LUI), lc1(V_RD_WRITE), lc1(V_IMM));
    const LC packed_query = pack_be(V_QUERY, C, LOG_M);
    eq_conditional(ins(I_ADD) + ins(I_ASSERT_ALIGN), packed_query, x + y);
    eq_conditional(ins(I_SUB), packed_query, x - y + lc1(-1, (int64_t)0xFFFFFFFFll + 1));
    const LC product = prod(V_PRODUCT, lc1(V_RS1), lc1(V_RS2));
    eq_conditional(ins(I_MUL) + ins(I_MULU) + ins(I_MULHU), packed_query, product);
    eq_conditional(ins(I_MOVSIGN) + ins(I_MOVE), packed_query, x);
    eq_conditional(op(F_ASSERT), lc1(V_OUT), one);
    eq_conditional(op(F_CONCAT), pack_be(V_CX, C, OPERAND_SIZE), x);
    eq_conditional(op(F_CONCAT), pack_be(V_CY, C, OPERAND_SIZE), y);
    const LC is_shift = ins(I_SLL) + ins(I_SRL) + ins(I_SRA);
    for (int i = 0; i < C; i++) {
        const LC rel = if_else(is_shift, lc1(V_CY + C - 1), lc1(V_CY + i), V_REL + i);
        eq_conditional(op(F_CONCAT), lc1(V_QUERY + i), lc1(V_CX + i, 1 << 8) + rel);
    }
    const LC w1 = prod(V_W1, lc1(V_BC_RD), op(F_WLO));
    eq_conditional(w1, lc1(V_RD_WRITE), lc1(V_OUT));
    const LC w2 = prod(V_W2, lc1(V_BC_RD), op(F_JUMP));
    eq_conditional(w2, lc1(V_ELF, 4) + lc1(-1, (int64_t)PC_START_ADDRESS), lc1(V_RD_WRITE));
    const LC npj = if_else(op(F_JUMP), lc1(V_OUT) + lc1(-1, 4), lc1(V_ELF, 4) + lc1(-1, (int64_t)PC_START_ADDRESS + 4) + lc1(V_OP + F_DNU, -4), V_NPJ);
    const LC sb = prod(V_SB, op(F_BRANCH), lc1(V_OUT));
    (void)if_else(sb, lc1(V_ELF, 4) + lc1(-1, (int64_t)PC_START_ADDRESS) + lc1(V_IMM), npj, V_NPC);
    // OffsetEqConstraint::new(cond, a, b): cond * (a - b) = 0
    s.add_cross(lc1(V_NPC), 0, lc1(V_ELF, 4) + lc1(-1, (int64_t)PC_START_ADDRESS), 1, lc1(V_ELF), 1);
    s.add_cross(lc1(V_BYTECODE_A), 1, lc1(V_BYTECODE_A) + lc1(-1, 1), 0, op(F_VIRTUAL), 0);
    s.finish(NUM_INPUTS);
}

static inline fe fr_i64(int64_t c) {
    fe m = Fr::from_u64((uint64_t)(c < 0 ? -c : c));
    return c < 0 ? Fr::neg(m) : m;
}

static inline size_t n_padding(size_t n) { return n / 8 > 1 ? n / 8 : 1; }

// the dealer's view of the 78 columns (oracle/pyjolt_r1cs.py synthetic_columns is the same generator)
static inline void build_clear(uint64_t seed, size_t n, std::vector<std::vector<fe>>& cols) {
    cols.assign(NUM_INPUTS, std::vector<fe>(n, Fr::zero()));
    auto sm = [&](uint64_t off, size_t t, int bits) { return (uint64_t)synthetic_small_host(seed + off, t, bits); };
    const size_t n_real = n - n_padding(n);
    auto F = [](uint64_t v) { return Fr::from_u64(v); };
    uint64_t elf = 1 + sm(22, 0, 16), bca = 5;
    static const int ALU[5] = {I_AND, I_OR, I_XOR, I_SLT, I_SLTU}, SHIFT[3] = {I_SLL, I_SRA, I_SRL}, MULS[3] = {I_MUL, I_MULU, I_MULHU},
                     BR[4] = {I_BEQ, I_BGE, I_BGEU, I_BNE}, ASRT[3] = {I_ASSERT_LTE, I_ASSERT_UREM, I_ASSERT_DIV0};
    enum { K_ALU, K_ALU_IMM, K_SHIFT, K_ADD, K_SUB, K_MUL, K_LOAD, K_STORE, K_LUI, K_JUMP, K_BRANCH, K_ASSERT, K_MOVE, K_ADDI, K_VIRT, K_ALIGN };
    for (size_t t = 0; t < n; t++) {
        if (t >= n_real) {
            cols[V_NPJ][t] = F(PC_START_ADDRESS + 4);
            cols[V_NPC][t] = F(PC_START_ADDRESS + 4);
            continue;
        }
        int kind = (int)(sm(11, t, 8) % 16);
        if (t == n_real - 1 && kind == K_VIRT) kind = K_ALU;
        int Fl[N_FLAGS] = {0};
        uint64_t rs1 = sm(13, t, 32), rs2 = sm(14, t, 32), imm_u = sm(15, t, 12), out = sm(16, t, 32);
        const uint64_t rd = sm(17, t, 6), regs = sm(18, t, 12), s_addr = sm(19, t, 20), s_misc = sm(20, t, 8), s_rdread = sm(21, t, 32), s_target = sm(22, t, 16);
        const uint64_t s_out = out, sel = sm(12, t, 8);
        uint64_t ram_addr = 0, ram_read = 0, ram_write = 0, rd_write = s_rdread ^ 0x5A5A5A5Aull;
        bool has_next = false;
        uint64_t nxt_elf = 0;
        int instr = 0;
        switch (kind) {
            case K_ALU: case K_ALU_IMM: case K_SHIFT:
                instr = kind == K_SHIFT ? SHIFT[sel % 3] : ALU[sel % 5];
                Fl[F_CONCAT] = 1;
                Fl[F_WLO] = 1;
                Fl[F_RIGHT_IMM] = kind == K_ALU_IMM;
                break;
            case K_ADD: case K_ADDI:
                instr = I_ADD;
                Fl[F_WLO] = 1;
                Fl[F_RIGHT_IMM] = kind == K_ADDI;
                break;
            case K_SUB:
                instr = I_SUB;
                Fl[F_WLO] = 1;
                break;
            case K_MUL:
                instr = MULS[sel % 3];
                Fl[F_WLO] = 1;
                break;
            case K_LOAD:
                instr = I_MOVE;
                Fl[F_LOAD] = 1;
                ram_addr = s_addr;
                rs1 = 4 * ram_addr + MEMORY_START - 4 * REGISTER_COUNT - imm_u;
                ram_read = ram_write = rd_write = s_out;
                break;
            case K_STORE:
                instr = I_MOVE;
                Fl[F_STORE] = 1;
                ram_addr = s_addr;
                rs1 = 4 * ram_addr + MEMORY_START - 4 * REGISTER_COUNT - imm_u;
                ram_read = s_rdread;
                ram_write = rs2;
                break;
            case K_LUI:
                instr = I_ADVICE;
                Fl[F_LUI] = 1;
                Fl[F_RIGHT_IMM] = 1;
                rd_write = imm_u;
                break;
            case K_JUMP:
                instr = I_ADD;
                Fl[F_JUMP] = Fl[F_LEFT_PC] = Fl[F_RIGHT_IMM] = 1;
                has_next = true;
                nxt_elf = 1 + s_target;
                break;
            case K_BRANCH:
                instr = BR[sel % 4];
                Fl[F_BRANCH] = Fl[F_CONCAT] = 1;
                out = s_misc & 1;
                if (out) {
                    has_next = true;
                    nxt_elf = 1 + s_target;
                }
                break;
            case K_ASSERT:
                instr = ASRT[sel % 3];
                Fl[F_ASSERT] = Fl[F_CONCAT] = 1;
                out = 1;
                break;
            case K_MOVE:
                instr = (sel & 1) ? I_MOVSIGN : I_MOVE;
                Fl[F_WLO] = 1;
                break;
            case K_VIRT:
                instr = I_ADVICE;
                Fl[F_VIRTUAL] = Fl[F_DNU] = Fl[F_WLO] = 1;
                break;
            default:  // K_ALIGN
                instr = I_ASSERT_ALIGN;
                Fl[F_ASSERT] = Fl[F_RIGHT_IMM] = 1;
                out = 1;
                break;
        }
        const uint64_t real_pc = 4 * elf + PC_START_ADDRESS - PC_NOOP_SHIFT;
        fe imm = F(imm_u);
        if (kind == K_JUMP) {
            out = 4 * nxt_elf + PC_START_ADDRESS - 4;
            imm = Fr::sub(F(out), F(real_pc));
        }
        if (kind == K_BRANCH && has_next) imm = Fr::sub(F(4 * nxt_elf), F(4 * elf));
        const uint64_t x_u = Fl[F_LEFT_PC] ? real_pc : rs1;
        const fe x = F(x_u), y = Fl[F_RIGHT_IMM] ? imm : F(rs2);
        const uint64_t y_u = Fl[F_RIGHT_IMM] ? imm_u : rs2;  // the operand as an integer where the chunks need one (never a jump)
        uint64_t cx[C] = {0}, cy[C] = {0}, q[C] = {0}, rel[C] = {0};
        if (Fl[F_CONCAT]) {
            const bool shift = instr == I_SLL || instr == I_SRA || instr == I_SRL;
            for (int i = 0; i < C; i++) {
                cx[i] = (x_u >> (8 * (C - 1 - i))) & 0xFF;
                cy[i] = (y_u >> (8 * (C - 1 - i))) & 0xFF;
            }
            for (int i = 0; i < C; i++) {
                rel[i] = shift ? cy[C - 1] : cy[i];
                q[i] = cx[i] * 256 + rel[i];
            }
        } else {
            uint64_t pq;
            if (kind == K_JUMP) pq = out;  // x + y = LookupOutput
            else if (instr == I_ADD || instr == I_ASSERT_ALIGN) pq = x_u + y_u;
            else if (instr == I_SUB) pq = x_u + ((uint64_t)1 << 32) - y_u;
            else if (instr == I_MUL || instr == I_MULU || instr == I_MULHU) pq = rs1 * rs2;
            else if (instr == I_MOVSIGN || instr == I_MOVE) pq = x_u;
            else pq = s_out ^ 0x1234ull;  // unconstrained by the R1CS
            for (int i = 0; i < C; i++) q[i] = (pq >> (LOG_M * (C - 1 - i))) & 0xFFFF;
        }
        if (kind == K_JUMP && rd != 0) rd_write = 4 * elf + PC_START_ADDRESS;
        if (Fl[F_WLO] && rd != 0) rd_write = out;
        const uint64_t w1 = rd * (uint64_t)Fl[F_WLO], w2 = rd * (uint64_t)Fl[F_JUMP];
        const uint64_t npj = Fl[F_JUMP] ? out + 4 : 4 * elf + PC_START_ADDRESS + 4 - 4 * (uint64_t)Fl[F_DNU];
        const uint64_t sb = (uint64_t)Fl[F_BRANCH] * out;
        if (!has_next) nxt_elf = elf + 1 - (uint64_t)Fl[F_DNU];
        const uint64_t npc = sb ? 4 * nxt_elf + PC_START_ADDRESS : npj;
        uint64_t bitflags = 0;
        for (int f = 0; f < N_FLAGS; f++) bitflags = 2 * bitflags + (uint64_t)Fl[f];
        for (int i = 0; i < N_INSTR; i++) bitflags = 2 * bitflags + (i == instr ? 1 : 0);
        cols[V_BYTECODE_A][t] = F(bca);
        cols[V_ELF][t] = F(elf);
        cols[V_BITFLAGS][t] = F(bitflags);
        cols[V_BC_RS1][t] = F(regs & 63);
        cols[V_BC_RS2][t] = F(regs >> 6);
        cols[V_BC_RD][t] = F(rd);
        cols[V_IMM][t] = imm;
        cols[V_RAM_ADDR][t] = F(ram_addr);
        cols[V_RS1][t] = F(rs1);
        cols[V_RS2][t] = F(rs2);
        cols[V_RD_READ][t] = F(s_rdread);
        cols[V_RAM_READ][t] = F(ram_read);
        cols[V_RD_WRITE][t] = F(rd_write);
        cols[V_RAM_WRITE][t] = F(ram_write);
        for (int i = 0; i < C; i++) {
            cols[V_QUERY + i][t] = F(q[i]);
            cols[V_CX + i][t] = F(cx[i]);
            cols[V_CY + i][t] = F(cy[i]);
            cols[V_REL + i][t] = F(rel[i]);
        }
        cols[V_OUT][t] = F(out);
        for (int f = 0; f < N_FLAGS; f++) cols[V_OP + f][t] = F((uint64_t)Fl[f]);
        cols[V_INSTR + instr][t] = Fr::one();
        cols[V_LEFT][t] = x;
        cols[V_RIGHT][t] = y;
        cols[V_PRODUCT][t] = Fr::mul(F(rs1), F(rs2));
        cols[V_W1][t] = F(w1);
        cols[V_W2][t] = F(w2);
        cols[V_NPJ][t] = F(npj);
        cols[V_SB][t] = F(sb);
        cols[V_NPC][t] = F(npc);
        bca = Fl[F_VIRTUAL] ? bca + 1 : 7 + (s_misc | (s_target << 8));
        elf = nxt_elf;
    }
}

}  // namespace jolt
