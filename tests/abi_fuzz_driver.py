"""Descriptor fuzz of the C ABI's host side (validation, geometry, workspace carving, job queues): run by
tests/test_abi_fuzz.py, in-process against libmedt_hip.so and in a subprocess against the AddressSanitizer build
(MEDT_LIB_OVERRIDE=libmedt_asan.so, LD_PRELOAD=<asan runtime>).  Only NULL / size arguments are passed where the ABI expects
device pointers, so every call ends in the validation layer (or, for size queries, in pure host arithmetic): nothing here
needs -- or touches -- a GPU.  Prints one JSON line with the call counts."""
import ctypes as C
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "medical-transformer_amd"), ROOT]
from medt_amd import _lib as L  # noqa: E402

EDGE = [0, 1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 4096, 65535, 65536,
        (1 << 20) + 1, (1 << 30), (1 << 31) - 1, -1, -2, -(1 << 31)]


def pick(rng, lo=None):
    r = rng.random()
    if r < 0.55:
        return rng.choice(EDGE)
    if r < 0.9:
        return rng.randint(0, 300)
    return rng.randint(-(1 << 31), (1 << 31) - 1)


def fval(rng):
    return rng.choice([1e-5, 0.1, 0.0, -1.0, 1e30, float("inf"), float("nan"), 1e-45])


def run(n, seed):
    lib = L.lib()
    rng = random.Random(seed)
    counts = {"axial_sizes": 0, "axial_ok": 0, "conv_sizes": 0, "conv_ok": 0, "block_sizes": 0, "entry_calls": 0, "queue_ops": 0}
    null = None
    for it in range(n):
        # ---- axial descriptors: wild, and near-valid (a valid shape with one field perturbed)
        if rng.random() < 0.5:
            vals = [pick(rng) for _ in range(10)]
        else:
            C_ = rng.choice([16, 32, 64, 128])
            Lh = rng.choice([2, 4, 8, 16, 32, 64, 128])
            vals = [rng.choice([1, 2, 4, 64]), C_, Lh, rng.choice([Lh, 2 * Lh, 4]), 8, rng.randint(0, 1), rng.randint(0, 1),
                    rng.choice([1, 2]), rng.randint(0, 1), rng.choice([1, 2, 16])]
            if rng.random() < 0.6:
                vals[rng.randrange(10)] = pick(rng)
        d = L.AxialDesc(*vals[:10], fval(rng) if rng.random() < 0.2 else 1e-5, fval(rng) if rng.random() < 0.2 else 0.1,
                        rng.choice([0, 1, 5, -1]), rng.choice([0, 1, 2, 3, -7]), rng.choice([0, 1, 2, -1]))
        ws = lib.medt_axial_workspace_bytes(C.byref(d))
        st = lib.medt_axial_stats_floats(C.byref(d))
        counts["axial_sizes"] += 1
        assert (ws == 0) == (st == 0), (vals, ws, st)
        if ws:
            counts["axial_ok"] += 1
            assert ws < (1 << 44) and st < (1 << 40), (vals, ws, st)          # no wrapped size arithmetic for accepted shapes
        p = L.AxialParams()
        sv = L.AxialSaved()
        g = L.AxialGrads()
        # every entry point with NULL tensors: refused with an error code and a message, never dereferenced
        for rc in (lib.medt_axial_layer_fwd(C.byref(d), C.byref(p), null, null, C.byref(sv), null, rng.choice([0, ws, 1 << 40]), null),
                   lib.medt_axial_layer_bwd(C.byref(d), C.byref(p), null, null, null, C.byref(sv), null, C.byref(g), null, ws, null),
                   lib.medt_axial_core_fwd(C.byref(d), C.byref(p), C.byref(sv), null, ws, null),
                   lib.medt_axial_core_stats(C.byref(d), C.byref(p), C.byref(sv), null, ws, null),
                   lib.medt_axial_core_bwd(C.byref(d), C.byref(p), C.byref(sv), null, null, ws, null)):
            counts["entry_calls"] += 1
            assert rc < 0, (vals, rc)
            assert lib.medt_last_error()
        # ---- convolution descriptors
        if rng.random() < 0.5:
            cv = [pick(rng) for _ in range(14)]
        else:
            cv = [rng.choice([1, 2, 4, 64]), rng.choice([3, 8, 16, 64, 128]), rng.choice([2, 4, 16, 32]), 0, rng.choice([2, 8, 64, 256]),
                  rng.choice([1, 3, 7]), rng.choice([1, 2]), rng.choice([0, 1, 3]), rng.randint(0, 1), rng.randint(0, 1),
                  rng.randint(0, 1), rng.randint(0, 1), rng.randint(0, 1), rng.choice([1, 2, 16])]
            cv[3] = cv[2]
            if rng.random() < 0.6:
                cv[rng.randrange(14)] = pick(rng)
        cd = L.ConvDesc(*cv, fval(rng) if rng.random() < 0.2 else 1e-5, 0.1, rng.choice([0, 1, 9]))
        cws = lib.medt_conv_workspace_bytes(C.byref(cd))
        lib.medt_conv_stats_floats(C.byref(cd))
        counts["conv_sizes"] += 1
        if cws:
            counts["conv_ok"] += 1
            assert cws < (1 << 46), (cv, cws)
        bn = L.BnPtrs()
        for rc in (lib.medt_conv_block_fwd(C.byref(cd), null, null, null, C.byref(bn), null, null, null, null, null, cws, null),
                   lib.medt_conv_block_bwd(C.byref(cd), null, null, C.byref(bn), null, null, null, null, null, null, null, null,
                                           null, null, null, null, cws, null)):
            counts["entry_calls"] += 1
            assert rc < 0, (cv, rc)
        # ---- block descriptors
        bv = [pick(rng) for _ in range(8)] if rng.random() < 0.6 else [64, 128, 64, 4, 4, 8, rng.randint(0, 1), 16]
        if rng.random() < 0.5:
            bv[rng.randrange(8)] = pick(rng)
        bd = L.BlockDesc(*bv, 1e-5, 0.1)
        bws = lib.medt_wopos_block_workspace_bytes(C.byref(bd))
        counts["block_sizes"] += 1
        assert bws < (1 << 30)
        bp, bs = L.BlockParams(), L.BlockSaved()
        assert lib.medt_wopos_block_fwd(C.byref(bd), C.byref(bp), null, null, C.byref(bs), null, bws, null) < 0
        assert lib.medt_wopos_block_fwd(None, None, None, None, None, None, 0, None) < 0
        bbw = lib.medt_wopos_block_bwd_workspace_bytes(C.byref(bd))       # 0 when MEDT_BLOCK_BWD=0 or not the fused shape
        assert bbw < (1 << 30)
        bg = L.BlockGrads()
        assert lib.medt_wopos_block_bwd(C.byref(bd), C.byref(bp), null, null, null, C.byref(bs), null, null, C.byref(bg), null,
                                        bbw, null) < 0
        assert lib.medt_wopos_block_bwd(None, None, None, None, None, None, None, None, None, None, 0, None) < 0
        counts["entry_calls"] += 4
        # ---- queues: create / bind / flush / discard / destroy in odd orders
        if it % 16 == 0:
            q = lib.medt_queue_create()
            stream = rng.choice([0, 1, 12345])
            assert lib.medt_queue_bind(q, stream) == 0
            assert lib.medt_queue_pending(q) == 0
            assert lib.medt_queue_flush(q, stream) == 0          # empty: no launch
            assert lib.medt_queue_discard(q) == 0
            if rng.random() < 0.5:
                assert lib.medt_queue_bind(None, stream) == 0    # unbind
            assert lib.medt_queue_destroy(q) == 0                # (destroy unbinds what is still bound)
            assert lib.medt_queue_flush(None, 0) < 0 and lib.medt_queue_discard(None) < 0 and lib.medt_queue_destroy(None) == 0
            assert lib.medt_queue_pending(None) == 0
            counts["queue_ops"] += 1
        # ---- the small entry points with NULL pointers / bad sizes
        assert lib.medt_up2x_relu_add_fwd(null, null, null, pick(rng), pick(rng), pick(rng), null) < 0
        assert lib.medt_up2x_relu_bwd(null, null, null, pick(rng), pick(rng), pick(rng), null) < 0
        assert lib.medt_patch_gather(null, null, 2, 3, 128, 32, 4, null) < 0
        assert lib.medt_logo_merge_fwd(null, null, null, 2, 3, 128, 32, 4, null) < 0
        assert lib.medt_logo_merge_bwd(null, null, null, 2, 3, 128, 32, 4, null) < 0
        assert lib.medt_ce_fwd(null, null, null, null, 2, 2, 64, -100, null) < 0
        assert lib.medt_ce_bwd(null, null, null, null, null, 2, 2, 64, -100, null) < 0
        assert lib.medt_adam_step(null, null, null, null, null, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, null) < 0
        assert lib.medt_relu_mask(null, null, null, 10, null) < 0
        assert lib.medt_seg_counts(null, null, null, 2, 2, 64, 0.5, null) < 0
        assert lib.medt_gate_mlp_fwd(*([null] * 9), 2, 16, 8, 8, 0, null) < 0
        assert lib.medt_gate_mlp_bwd(*([null] * 13), 2, 16, 8, 8, 0, null) < 0
        counts["entry_calls"] += 12
    return counts


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print(json.dumps(run(n, seed)))
