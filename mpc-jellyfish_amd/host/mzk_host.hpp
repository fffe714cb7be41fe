// mzk_host.hpp -- C++ host side of the jf-plonk prover above the C ABI of include/mzk.h (no HIP, no Python):
// what a compiled host (the reference is Rust) does around the device calls.  Mirrors, with the reference's names,
//
//   jf_utils::test_rng / F::rand / DensePolynomial::rand      utilities/src/lib.rs:62-70 [+ upstream ark-ff, rand_chacha]
//   compute_coset_representatives                            relation/src/constants.rs:30-80
//   StandardTranscript over merlin                           plonk/src/transcript/standard.rs:16-46, transcript/mod.rs:45-214
//   gen_circuit_for_bench + finalize_for_arithmetization     plonk/benches/bench.rs:29-46, relation/src/constraint_system.rs:195-225, 743-778, 966-999
//   PlonkKzgSnark::{preprocess, prove}                       plonk/src/proof_system/snark.rs:201-469, 529-651
//   Prover::run_*_round, compute_*                            plonk/src/proof_system/prover.rs:72-509, 902-1122
//   Proof::serialize_compressed                              plonk/src/proof_system/structs.rs:59-84, 208-222, 440-450, 496-541
//
// Every field vector stays in HBM between the rounds; the host holds scalars (challenges, evaluations, blinders),
// commitments and the transcript.  The Python package (mpc-jellyfish_amd/*.py) is the same logic for the test-suite;
// tests/test_host_cpp_gpu.py checks that both produce the same proof bytes.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mzk.h"
#include "../csrc/hostfp.hpp"

namespace mzk_host {

using mzk::Fp64;

inline void check(int32_t rc, const char* where) {
    if (rc != MZK_OK) throw std::runtime_error(std::string(where) + ": " + mzk_strerror(rc) + ": " + mzk_last_error());
}

// host field helpers on top of Fp64 (from_u64, pow_u64, inv, canonical, root_of_unity): csrc/hostfp.hpp -- the library's own prover rounds
// (csrc/prover.hip) use the same ones
using mzk::h64::canonical;
using mzk::h64::from_u64;
using mzk::h64::inv;
using mzk::h64::pow_u64;
using mzk::h64::root_of_unity;

// ---- rand_chacha ChaCha{8,12,20}Rng and ark-ff's Fp::rand ------------------------------------------------
struct ChaChaRng {
    uint32_t key[8];
    int rounds;
    uint64_t counter = 0;
    uint32_t buf[64];
    int index = 64;
    ChaChaRng(const uint8_t seed[32], int rounds_) : rounds(rounds_) { std::memcpy(key, seed, 32); }
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void block(uint32_t* out) {
        uint32_t init[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                             (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t x[16];
        std::memcpy(x, init, sizeof x);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int i = 0; i < rounds / 2; i++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) out[i] = x[i] + init[i];
        counter++;
    }
    void refill() {
        for (int b = 0; b < 4; b++) block(buf + 16 * b);          // rand_chacha produces four blocks per refill
        index = 0;
    }
    uint64_t next_u64() {                                         // rand_core::block::BlockRng (the index stays even here)
        if (index >= 64) refill();
        const uint64_t v = (uint64_t)buf[index] | ((uint64_t)buf[index + 1] << 32);
        index += 2;
        return v;
    }
};
inline ChaChaRng test_rng() {                                     // utilities/src/lib.rs:62-70: StdRng = ChaCha12
    const uint8_t seed[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0};
    return ChaChaRng(seed, 12);
}
// `Fr::rand`: N u64 limbs, top bits shaved, rejection below the modulus; the limbs ARE the Montgomery representation
template <class P>
Fp64<P> fr_rand(ChaChaRng& rng) {
    constexpr int N = Fp64<P>::N;
    const int shave = 64 * N - P::BITS;
    for (;;) {
        Fp64<P> v;
        for (int i = 0; i < N; i++) v.l[i] = rng.next_u64();
        v.l[N - 1] &= ~0ull >> shave;
        if (!Fp64<P>::geq_mod(v.l)) return v;
    }
}
template <class P>
std::vector<Fp64<P>> compute_coset_representatives(int num_wire_types, uint64_t coset_size) {     // relation/src/constants.rs:30-80
    const uint8_t zero_seed[32] = {0};
    ChaChaRng rng(zero_seed, 20);
    std::vector<Fp64<P>> ks{Fp64<P>::one()}, pows{Fp64<P>::one()};
    for (int i = 1; i < num_wire_types; i++) {
        for (;;) {
            const Fp64<P> k = fr_rand<P>(rng), p = pow_u64(k, coset_size);
            if (std::find(pows.begin(), pows.end(), p) == pows.end()) { ks.push_back(k); pows.push_back(p); break; }
        }
    }
    return ks;
}

// ---- merlin (STROBE-128 over Keccak-f[1600]) ------------------------------------------------------------
struct Strobe128 {
    static constexpr int R = 166;
    enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };
    uint8_t st[200];
    int pos = 0, pos_begin = 0, cur_flags = 0;
    explicit Strobe128(const std::string& protocol_label) {
        std::memset(st, 0, sizeof st);
        const uint8_t head[6] = {1, R + 2, 1, 0, 1, 96};
        std::memcpy(st, head, 6);
        std::memcpy(st + 6, "STROBEv1.0.2", 12);
        check(mzk_keccak_f1600(st), "mzk_keccak_f1600");
        meta_ad(reinterpret_cast<const uint8_t*>(protocol_label.data()), protocol_label.size(), false);
    }
    void run_f() {
        st[pos] ^= (uint8_t)pos_begin;
        st[pos + 1] ^= 0x04;
        st[R + 1] ^= 0x80;
        check(mzk_keccak_f1600(st), "mzk_keccak_f1600");
        pos = pos_begin = 0;
    }
    void absorb(const uint8_t* d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            st[pos++] ^= d[i];
            if (pos == R) run_f();
        }
    }
    void squeeze(uint8_t* d, size_t n) {
        for (size_t i = 0; i < n; i++) {
            d[i] = st[pos];
            st[pos++] = 0;
            if (pos == R) run_f();
        }
    }
    void begin_op(int flags, bool more) {
        if (more) return;                                        // continuation of the same operation
        const int old_begin = pos_begin;
        pos_begin = pos + 1;
        cur_flags = flags;
        const uint8_t hdr[2] = {(uint8_t)old_begin, (uint8_t)flags};
        absorb(hdr, 2);
        if ((flags & (FLAG_C | FLAG_K)) && pos != 0) run_f();
    }
    void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
    void ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
    void prf(uint8_t* d, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(d, n); }
};
struct MerlinTranscript {
    Strobe128 s;
    explicit MerlinTranscript(const std::string& label) : s("Merlin v1.0") { append_message("dom-sep", reinterpret_cast<const uint8_t*>(label.data()), label.size()); }
    static void le32(uint8_t* o, uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); }
    void append_message(const std::string& label, const uint8_t* msg, size_t n) {
        uint8_t len[4];
        le32(len, (uint32_t)n);
        s.meta_ad(reinterpret_cast<const uint8_t*>(label.data()), label.size(), false);
        s.meta_ad(len, 4, true);
        s.ad(msg, n, false);
    }
    void challenge_bytes(const std::string& label, uint8_t* out, size_t n) {
        uint8_t len[4];
        le32(len, (uint32_t)n);
        s.meta_ad(reinterpret_cast<const uint8_t*>(label.data()), label.size(), false);
        s.meta_ad(len, 4, true);
        s.prf(out, n, false);
    }
};

// ---- curve bundles ------------------------------------------------------------------------------------
struct Bls12_381 {
    using Fr = mzk::BlsFr;
    using Fq = mzk::BlsFq;
    static constexpr int ID = MZK_CURVE_BLS12_381;
    static constexpr int G1_BYTES = 48;
};
struct Bn254 {
    using Fr = mzk::BnFr;
    using Fq = mzk::BnFq;
    static constexpr int ID = MZK_CURVE_BN254;
    static constexpr int G1_BYTES = 32;
};

template <class C>
struct Encoding {
    using Fr = Fp64<typename C::Fr>;
    using Fq = Fp64<typename C::Fq>;
    static constexpr int QL = Fq::N;                                  // u64 limbs of Fq
    using Affine = std::array<uint64_t, 2 * Fq::N>;                  // x || y, Montgomery; all zero = infinity

    static void fr_bytes(const Fr& v, uint8_t out[32]) {              // ark-serialize: 32 bytes little-endian canonical
        const auto c = canonical(v);
        std::memcpy(out, c.data(), 32);
    }
    // ark-serialize compressed G1: BLS12-381 in the zcash form (48 bytes big-endian, flags in the top three bits),
    // BN254 little-endian with y-sign in bit 7 and infinity in bit 6 of the last byte
    static void g1_bytes(const Affine& p, uint8_t* out) {
        bool inf = true;
        for (uint64_t w : p) inf &= (w == 0);
        std::memset(out, 0, C::G1_BYTES);
        if (C::ID == MZK_CURVE_BLS12_381) {
            if (inf) { out[0] = 0xC0; return; }
        } else if (inf) { out[C::G1_BYTES - 1] = 0x40; return; }
        Fq x, y;
        std::memcpy(x.l, p.data(), sizeof x.l);
        std::memcpy(y.l, p.data() + QL, sizeof y.l);
        const auto xc = canonical(x), yc = canonical(y), nyc = canonical(mzk::neg(y));
        bool y_big = false;                                           // y > -y  <=>  y > (q - 1) / 2
        for (int i = QL - 1; i >= 0; i--)
            if (yc[i] != nyc[i]) { y_big = yc[i] > nyc[i]; break; }
        if (C::ID == MZK_CURVE_BLS12_381) {
            for (int i = 0; i < QL; i++)
                for (int b = 0; b < 8; b++) out[C::G1_BYTES - 1 - (8 * i + b)] = (uint8_t)(xc[i] >> (8 * b));
            out[0] |= 0x80 | (y_big ? 0x20 : 0);
        } else {
            std::memcpy(out, xc.data(), C::G1_BYTES);
            if (y_big) out[C::G1_BYTES - 1] |= 0x80;
        }
    }
};

// plonk/src/transcript/standard.rs over merlin; labels of transcript/mod.rs
template <class C>
struct StandardTranscript {
    using E = Encoding<C>;
    using Fr = typename E::Fr;
    MerlinTranscript t;
    explicit StandardTranscript(const char* label = "PlonkProof") : t(label) {}
    void append_message(const std::string& label, const uint8_t* m, size_t n) { t.append_message(label, m, n); }
    void append_u64(const std::string& label, uint64_t v) { uint8_t b[8]; std::memcpy(b, &v, 8); t.append_message(label, b, 8); }
    void append_u32(const std::string& label, uint32_t v) { uint8_t b[4]; std::memcpy(b, &v, 4); t.append_message(label, b, 4); }
    void append_fr(const std::string& label, const Fr& v) { uint8_t b[32]; E::fr_bytes(v, b); t.append_message(label, b, 32); }
    void append_commitment(const std::string& label, const typename E::Affine& p) { uint8_t b[48]; E::g1_bytes(p, b); t.append_message(label, b, C::G1_BYTES); }
    Fr get_and_append_challenge(const std::string& label) {          // 64 squeezed bytes, little-endian, reduced mod r; then re-absorbed
        uint8_t buf[64];
        t.challenge_bytes(label, buf, 64);
        // from_le_bytes_mod_order: value = lo + hi * 2^256
        Fp64<typename C::Fr> lo = Fr::zero(), hi = Fr::zero();
        uint64_t w[8];
        std::memcpy(w, buf, 64);
        // reduce each 256-bit half: load as an integer < 2^256 and multiply by R2 (Montgomery conversion reduces mod r)
        Fr r2;
        for (int i = 0; i < 4; i++) { lo.l[i] = w[i]; hi.l[i] = w[4 + i]; r2.l[i] = Fr::c64(C::Fr::R2, i); }
        // the halves are raw 256-bit integers, up to 2.2 r (BLS12-381) / 5.3 r (BN254): bring them below r first -- the Montgomery
        // product (no-carry CIOS) is only defined for reduced operands and drops a carry for larger ones
        while (Fr::geq_mod(lo.l)) Fr::sub_mod(lo.l);
        while (Fr::geq_mod(hi.l)) Fr::sub_mod(hi.l);
        const Fr lo_m = lo * r2, hi_m = hi * r2;                      // (x mod r) in Montgomery form
        Fr two256 = Fr::one();                                        // Montgomery image of 1 is R = 2^256 mod r, i.e. the integer 2^256: as a field element it is r2 * 1
        two256 = r2;                                                  // value(r2 as Montgomery) = R2 / R = R = 2^256 mod r
        const Fr c = lo_m + hi_m * two256;
        append_fr(label, c);
        return c;
    }
};

}  // namespace mzk_host
