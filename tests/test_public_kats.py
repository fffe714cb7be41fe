"""CPU suite: PUBLIC third-party known answers, computed by the oracle and by the library's host-only entry points (VERDICT r2 #5a).
None of these values comes from this repository or from the reference: a wrong recollection of them cannot agree with an
independent computation by accident, and an agreement pins the group law / the permutation independently of who wrote the code.

    BN254 (alt_bn128) [2]G1, G1 = (1, 2)            EIP-196 precompile test vectors (ecAdd / ecMul of the generator)
    BLS12-381 [2]G1                                   EIP-2537 / IETF pairing-friendly-curves draft test vectors; zkcrypto `bls12_381` doubling test
    ChaCha20 zero-key blocks 0 and 1                  RFC 7539 Appendix A.1 test vectors #1 and #2
    SHA3-256(""), Keccak-256("")                      FIPS 202 / the Ethereum empty-hash constant
The GPU twins (the device MSM and the device SRS generator) are in tests/test_public_kats_gpu.py."""
import ctypes as C
import struct

import numpy as np

import pyref as P
import pyref_fs as FS
import pyref_rng as RNG

BN254_2G = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
            9918110051302171585080402603319702774565515993150576347155970296011118125764)
BLS_G = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
         0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
BLS_2G = (0x0572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e,
          0x166a9d8cabc673a322fda673779d8e3822ba3ecb8670e461f73bb9021d5fd76a4c56d9d4cd16bd1bba86881979749d28)
PUBLIC_2G = {0: BLS_2G, 1: BN254_2G}
CHACHA20_ZERO_KEY_BLOCK0 = "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586"
CHACHA20_ZERO_KEY_BLOCK1 = "9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed29b721769ce64e43d57133b074d839d531ed1f28510afb45ace10a1f4b794d6f"
SHA3_256_EMPTY = "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
KECCAK_256_EMPTY = "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"


def test_oracle_group_law_against_public_doublings(cref):
    assert (P.BN254.gx, P.BN254.gy) == (1, 2) and (P.BLS12_381.gx, P.BLS12_381.gy) == BLS_G
    for cid, pc in sorted(P.CURVES.items()):
        G = P.g1_gen(pc)
        assert P.g1_mul(pc, 2, G) == PUBLIC_2G[cid]                              # big-int definitions (oracle/pyref.py)
        assert P.g1_add(pc, G, G) == PUBLIC_2G[cid]
        # the C oracle (oracle/cpu_ref.c): fixed-base multiple, and its Pippenger on the one pair (2, G)
        L = cref.fq_limbs(cid)
        two_g = cref.fq_convert(cid, cref.g1_mul_gen(cid, 2).reshape(2, L), False)
        assert tuple(cref.limbs_to_ints(two_g)) == PUBLIC_2G[cid]
        g_xy = cref.g1_mul_gen(cid, 1).reshape(1, 2 * L)
        jac = cref.msm(cid, g_xy, cref.ints_to_limbs([2], 4))
        aff = cref.fq_convert(cid, cref.jac_to_affine(cid, jac).reshape(2, L), False)
        assert tuple(cref.limbs_to_ints(aff)) == PUBLIC_2G[cid]


def test_library_host_group_law_against_public_doublings(mj):
    """mzk_g1_sum_jacobian / mzk_g1_jacobian_to_affine (host-only: what combines the per-GPU partial commitments): G + G."""
    L = mj.load()
    for cid, pc in sorted(P.CURVES.items()):
        c = mj.params.CURVES[cid]
        one = mj.params.fq_to_mont(c, [1])[0]
        g = np.concatenate([mj.params.fq_to_mont(c, [pc.gx])[0], mj.params.fq_to_mont(c, [pc.gy])[0], one])          # Jacobian (x, y, 1)
        two = np.ascontiguousarray(np.stack([g, g]))
        out = np.zeros(3 * c.fq_limbs, dtype=np.uint64)
        assert L.mzk_g1_sum_jacobian(cid, C.c_void_p(two.ctypes.data), 2, C.c_void_p(out.ctypes.data)) == 0
        aff = np.zeros(2 * c.fq_limbs, dtype=np.uint64)
        assert L.mzk_g1_jacobian_to_affine(cid, C.c_void_p(out.ctypes.data), 1, C.c_void_p(aff.ctypes.data)) == 0
        assert tuple(mj.params.fq_from_mont(c, aff)) == PUBLIC_2G[cid]


def test_chacha20_rfc7539_appendix_a1(mj):
    L = mj.load()
    for counter, kat in ((0, CHACHA20_ZERO_KEY_BLOCK0), (1, CHACHA20_ZERO_KEY_BLOCK1)):
        assert struct.pack("<16I", *RNG.chacha_words(bytes(32), counter, 20)).hex() == kat
        out, zero = (C.c_uint32 * 16)(), (C.c_uint32 * 8)()
        assert L.mzk_chacha_blocks(C.addressof(zero), counter, 20, 1, C.addressof(out)) == 0
        assert struct.pack("<16I", *out).hex() == kat


def test_fips202_empty_message_hashes(mj):
    assert FS.sha3_256(b"").hex() == SHA3_256_EMPTY and FS.keccak256(b"").hex() == KECCAK_256_EMPTY
    for perm in (mj.transcript.keccak_f1600, mj.transcript.keccak_f1600_py):       # the library's exported permutation and its Python twin
        for domain_byte, kat in ((0x06, SHA3_256_EMPTY), (0x01, KECCAK_256_EMPTY)):
            st = bytearray(200)
            st[0] ^= domain_byte
            st[135] ^= 0x80                                                       # rate 136 bytes
            perm(st)
            assert bytes(st[:32]).hex() == kat
