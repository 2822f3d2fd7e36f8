"""Serialised backbone programs: what a host WITHOUT Python needs to evaluate the network through the C ABI.

``Program`` (program.py) records the forward of ``NCSNpp_v2`` (fdbm/backbones/ncsnpp_v2.py:241-401) for one
(B, F, T, dtype) as ~150 ``fdbm_op`` entries holding raw device pointers.  ``export_program`` rewrites every pointer
as (region, offset) - region 0: ONE caller-provided workspace (activations, statistics arena, scratch, the static
I/O buffers), region 1: ONE flat weight blob in the device formats the kernels read (packed / fragment-major conv
weights, GroupNorm and bias vectors, the time-embedding matrices) - and returns two byte strings:

    program  header | fdbm_op[n_ops] | fdbm_conv_args[n_conv]     (include/fdbm_hip.h, `fdbm_program_header`)
    weights  the blob (upload it once, any 256-byte aligned device address)

C side (csrc/runtime.cpp): fdbm_program_workspace_bytes / fdbm_program_weights_bytes size the two buffers,
fdbm_ncsnpp_create_from_program(program, n, weights_dev, workspace_dev, workspace_bytes) relocates and returns the
same context fdbm_ncsnpp_create gives; fdbm_ncsnpp_forward runs it.  examples/host_cpp/run_program.cpp is a complete
host in C++ (no Python at run time); tools/export_program.py writes the two files.
"""
import ctypes
import struct

import torch

from . import hip

MAGIC = 0x474F5250424D4446        # "FDMBPROG" little endian
VERSION = 1
PTR_TAG = 1 << 62
_CONV_PTR_FIELDS = ("w", "bias", "tbias", "res", "out", "workspace", "gn_sums", "gn_gamma", "gn_beta",
                    "comb_pyr", "comb_w", "comb_b", "stat_out", "w_frag", "acc_ws", "res_up2x")
_ALIGN = 256


def _weight_tensors(net):
    out = []
    for v in net.w.values():
        out += [t for t in v.values() if torch.is_tensor(t)]
    out += [t for (_, t) in getattr(net, "_frag", {}).values()]
    out += [v[1] for v in getattr(net, "_split", {}).values()]            # pre-split weights of the split-precision mode
    out += [net.fourier_w, net.lin1_w, net.lin1_b, net.lin2_w, net.lin2_b, net.dense_w, net.dense_b]
    return out


def export_program(prog):
    """-> (program bytes, weights bytes) of a built ``Program``."""
    net = prog.net
    # ---- regions: every device allocation the ops may point into -----------------------------------------
    ws_allocs = list(prog.pool.all) + [prog.arena, prog.x_in, prog.y_in, prog.t_in, prog.s_out] + list(prog.keep)
    w_allocs = _weight_tensors(net)
    table = []                                  # (base, nbytes, region, offset)

    def lay(allocs, region):
        off, seen = 0, set()
        for t in allocs:
            base = t.data_ptr()
            if base in seen:
                continue
            seen.add(base)
            nb = t.numel() * t.element_size()
            table.append((base, nb, region, off))
            off += (nb + _ALIGN - 1) // _ALIGN * _ALIGN
        return off
    ws_bytes = lay(ws_allocs, 0)
    w_bytes = lay(w_allocs, 1)
    table.sort()

    def in_table(p):
        import bisect
        i = bisect.bisect_right(table, (p, 1 << 62, 9, 0)) - 1
        return i >= 0 and table[i][0] <= p < table[i][0] + table[i][1] + _ALIGN

    def reloc(p):
        if p == 0:
            return 0
        import bisect
        i = bisect.bisect_right(table, (p, 1 << 62, 9, 0)) - 1
        assert i >= 0, hex(p)
        base, nb, region, off = table[i]
        assert base <= p < base + nb + _ALIGN, f"pointer {hex(p)} is not inside any known allocation"
        return PTR_TAG | (region << 60) | (off + (p - base))

    # ---- ops (OP_CONV: iarg[0] = index into the conv-args table) ------------------------------------------
    ops = (hip.Op * prog.n_ops)()
    ctypes.memmove(ops, prog.op_array, ctypes.sizeof(ops))
    conv_index = {ctypes.addressof(ca): i for i, ca in enumerate(prog.keep_conv)}
    for i in range(prog.n_ops):
        if ops[i].opcode == hip.OP_CONV:
            ops[i].iarg[0] = conv_index[ops[i].iarg[0]]
            continue
        for j in range(24):
            v = ops[i].iarg[j]
            # A device address is recognised by MEMBERSHIP in one of the program's allocations, not by its magnitude
            # (ADVICE r2): a value inside an allocation is relocated; counts, shapes and strides (all < 2^32) lie far below
            # any device mapping; anything else that large is an address this exporter does not know - refuse it.
            if in_table(v):
                ops[i].iarg[j] = reloc(v)
            elif v >= (1 << 32):
                raise ValueError(f"op {i} (opcode {ops[i].opcode}) argument {j} = {hex(v)} is neither a count nor inside a known allocation")
    convs = (hip.ConvArgs * len(prog.keep_conv))()
    for i, ca in enumerate(prog.keep_conv):
        ctypes.memmove(ctypes.byref(convs[i]), ctypes.byref(ca), ctypes.sizeof(hip.ConvArgs))
        for s in range(hip.MAX_SEG):
            convs[i].seg[s].src = reloc(ca.seg[s].src or 0)
            convs[i].gn_seg_sums[s] = reloc(ca.gn_seg_sums[s] or 0)
        for f in _CONV_PTR_FIELDS:
            setattr(convs[i], f, reloc(getattr(ca, f) or 0))
    header = struct.pack("<QIIIIqqiiiiqqqqq", MAGIC, VERSION, prog.n_ops, len(prog.keep_conv), ctypes.sizeof(hip.ConvArgs),
                         ws_bytes, w_bytes, prog.B, prog.F, prog.T, prog.dtc,
                         reloc(prog.x_in.data_ptr()), reloc(prog.y_in.data_ptr()), reloc(prog.t_in.data_ptr()),
                         reloc(prog.s_out.data_ptr()), prog.x_in.numel())
    header = header.ljust(128, b"\0")
    program = header + bytes(ops) + bytes(convs)
    # ---- weight blob ------------------------------------------------------------------------------------------
    blob = bytearray(w_bytes)
    for base, nb, region, off in table:
        if region == 1:
            t = next(t for t in w_allocs if t.data_ptr() == base)
            blob[off:off + nb] = t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes()
    return program, bytes(blob)


def load_program(program, weights_dev, workspace_dev):
    """Python binding of the C loader (tests): -> context handle; the two tensors must outlive it."""
    L = hip.lib()
    n = len(program)
    buf = ctypes.create_string_buffer(program, n)
    need_ws = L.fdbm_program_workspace_bytes(buf, n)
    need_w = L.fdbm_program_weights_bytes(buf, n)
    assert need_ws >= 0 and need_w >= 0, L.fdbm_last_error().decode()
    assert workspace_dev.numel() * workspace_dev.element_size() >= need_ws
    assert weights_dev.numel() * weights_dev.element_size() >= need_w
    ctx = L.fdbm_ncsnpp_create_from_program(buf, n, weights_dev.data_ptr(), workspace_dev.data_ptr(),
                                            workspace_dev.numel() * workspace_dev.element_size())
    if not ctx:
        raise RuntimeError("fdbm_ncsnpp_create_from_program failed: " + L.fdbm_last_error().decode())
    return ctx
