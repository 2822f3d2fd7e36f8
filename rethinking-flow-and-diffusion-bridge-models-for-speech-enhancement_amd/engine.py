"""Whole-sampler HIP graphs: N x { backbone program ; fused state update } recorded once per
(shape, sampler, N, path) and replayed per batch of utterances.

The reference's inner loop (fdbm/bridge.py:73-85, :96-111) launches ~1.1k framework ops
plus ~45 tiny scalar kernels per step from Python.  Here one replay of one graph runs
all N steps: first the time-embedding rows of all N steps (the Dense_0 projections of every res-block: a function of
the time grid only, hoisted out of the step loop), then per step the recorded backbone program between its input packing
and its output conv (program.py, run_core) and ONE fdbm_step_boundary launch: score from the output pyramid, the state
update, the next evaluation's packed input, its zeroed statistics arena and its time-embedding rows.
The per-step weights come from the host table (bridge.ei_weight_table) uploaded once.
"""
import os

import torch

from . import hip


class SamplerGraph:
    def __init__(self, net, bridge, kind, B, F, T, in_kernel_noise=False):
        self.net, self.kind, self.N = net, kind, bridge.N
        # sde with the library's counter-based generator: every step's noise is generated inside its step boundary
        # (fdbm_step_boundary_rng, draw 1 + i); self.rng = the device words {seed_lo, seed_hi, draw_base} it reads
        self.rng = None
        self.prog = net.program(B, F, T)
        dev = net.device
        table, t_model = bridge.ei_weight_table("ode" if kind == "ode" else "sde", B)
        self.table = table.to(dev).contiguous()                       # [N,3,B]
        # model time of each step as log t, evaluated on the host (see fdbm_temb)
        self.t_tab = hip.log_time(t_model[:, None] * torch.ones(1, B)).to(dev).contiguous()   # [N,B]
        self.z = None
        if kind == "sde" and in_kernel_noise:
            from .bridge import rng_state
            self.rng = rng_state(0, dev)
        elif kind == "sde":
            self.z = torch.zeros(self.N, B, 1, F, T, dtype=torch.complex64, device=dev)
        self.graph = None
        self.key = self._bridge_key(bridge)
        self.dense_tab = self.dense_bufs = None      # [N*B, dense_rows]: the Dense_0 rows of every step's t (_steps)

    @staticmethod
    def _bridge_key(bridge):
        p = bridge.path
        return (type(p).__name__, bridge.N, bridge.start_time, bridge.end_time,
                tuple(sorted((k, v) for k, v in vars(p).items() if isinstance(v, (int, float, str)))))

    def _steps(self):
        prog = self.prog
        B, F, T = prog.B, prog.F, prog.T
        R = self.net.dense_rows
        p = hip.ptr
        # the time-embedding chain (2 + 1 launches per evaluation in the reference's loop) is a function of the time grid
        # only: hoisted out of the step loop - evaluated HERE for all N steps, once per sampler call, inside the graph
        self.dense_tab = prog.dense_table(self.t_tab, self.dense_bufs)
        ba = prog.boundary_args()

        def boundary(i_done, i_next):
            """Between evaluation i_done (None: none yet) and evaluation i_next (None: none left): score -> state update ->
            next network input, arena zeroing and time-embedding rows, ONE launch (fdbm_step_boundary)."""
            upd = i_done is not None
            nxt = i_next is not None
            w = self.table[i_done] if upd else None
            if upd and self.rng is not None:
                name, third_args = "fdbm_step_boundary_rng", (p(self.rng), 1 + i_done)
            else:
                third = (prog.y_in if self.kind == "ode" else self.z[i_done]) if upd else None
                name, third_args = "fdbm_step_boundary", (p(third),)
            hip.call(name, p(prog.x_in), p(prog.y_in), *third_args,
                     ba["pyramid"] if upd else 0, ba["out_w"] if upd else 0, ba["out_b"] if upd else 0,
                     p(w[0]) if upd else 0, p(w[1]) if upd else 0, p(w[2]) if upd else 0,
                     ba["packed"] if nxt else 0, ba["arena"] if nxt else 0, ba["arena_bytes"] if nxt else 0,
                     p(prog.dense_out) if nxt else 0, p(self.dense_tab[i_next * B]) if nxt else 0, B * R if nxt else 0,
                     B, F, ba["Fn"], T)

        if os.environ.get("FDBM_STEP_BOUNDARY", "1") == "0":          # A/B (tools/ab.py step): the five separate launches per step
            n = prog.x_in[0].numel()
            for i in range(self.N):
                hip.call("fdbm_copy_f32", p(prog.dense_out), p(self.dense_tab[i * B]), B * R)
                prog.run_body()
                assert self.rng is None, "FDBM_STEP_BOUNDARY=0 (A/B) is built for injected noise"
                third = prog.y_in if self.kind == "ode" else self.z[i]
                w = self.table[i]
                hip.call("fdbm_bridge_update", p(prog.x_in), p(prog.x_in), p(prog.s_out), p(third), p(w[0]), p(w[1]), p(w[2]), B, n)
            return
        boundary(None, 0)
        for i in range(self.N):
            prog.run_core()
            boundary(i, i + 1 if i + 1 < self.N else None)

    def capture(self):
        self.dense_bufs = self.prog.dense_table_buffers(self.t_tab.numel())
        side = torch.cuda.Stream(device=self.net.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture (lazy module loads, attributes)
            self._steps()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._steps()
        self.graph = g

    def run(self, y, x0, step_noise=None, seed=None):
        """y, x0: complex64 [B,1,F,T] device tensors; returns the final state (new tensor)."""
        prog = self.prog
        prog.y_in.copy_(y)
        prog.x_in.copy_(x0)
        if self.rng is not None:
            from .bridge import rng_state
            self.rng.copy_(rng_state(seed, "cpu"), non_blocking=True)       # 16 bytes: this call's seed; the draws are made in the graph
        elif self.kind == "sde":
            for i in range(self.N):
                self.z[i].copy_(step_noise())
        if self.graph is None:
            self.capture()
            prog.y_in.copy_(y)
            prog.x_in.copy_(x0)
        self.graph.replay()
        return prog.x_in.clone()


class PcGraph:
    """Predictor-corrector sampler (fdbm/bridge.py:142-166 with EulerMaruyamaPredictor, fdbm/util/predictors.py:40-51,
    and AnnealedLangevinDynamics / LangevinCorrector / none, fdbm/util/correctors.py:37-81) as ONE HIP graph: per grid
    point {corrector_steps x [backbone ; corrector move] ; backbone ; predictor move}.  Every scalar of ald / the
    predictor is a host table uploaded once; the Langevin corrector's norm-ratio step size is computed by
    fdbm_langevin_step on the device.  Noise of all steps is drawn on the host in the reference's call order and
    uploaded before the replay."""

    def __init__(self, net, bridge, B, F, T, predictor_name, corrector_name, snr, n_steps, in_kernel_noise=False):
        from .paths import ProbabilityPathSB  # noqa: F401  (sde_weights lives on the path)
        self.net, self.N, self.B = net, bridge.N, B
        self.rng = None
        self.pred, self.corr, self.snr = predictor_name, corrector_name, float(snr)
        self.n_steps = n_steps if corrector_name != "none" else 0
        self.prog = net.program(B, F, T)
        dev = net.device
        ts = bridge.time_grid(self.N)
        path = bridge.path
        ctab, ptab, dts = [], [], []
        for i in range(self.N):
            vt = torch.ones(B) * ts[i]
            a_t, b_t, sig = path.path_param(vt)
            den = sig ** 2 + 1e-8
            std = bridge._std(vt)
            step = (self.snr * std) ** 2 * 2
            ctab.append(torch.stack([a_t, b_t, den, step, torch.sqrt(step * 2)]).to(torch.float32))
            ptab.append(torch.stack([w.to(torch.float32) for w in path.sde_weights(vt)]))
            stepsize = ts[i] - ts[i + 1] if i != self.N - 1 else ts[-1]
            dts.append(-float(stepsize))
        self.ctab = torch.stack(ctab).to(dev).contiguous()         # [N,5,B]
        self.ptab = torch.stack(ptab).to(dev).contiguous()         # [N,4,B]
        self.dts = dts
        self.t_tab = hip.log_time(ts[:, None] * torch.ones(1, B)).to(dev).contiguous()
        z = lambda *lead: torch.zeros(*lead, B, 1, F, T, dtype=torch.complex64, device=dev)
        if in_kernel_noise:
            # the library's counter-based generator: draws numbered in the reference's call order (per grid point the
            # corrector's, then the predictor's), each generated inside the move that consumes it; only the Langevin
            # corrector, whose step size needs the draw's norm first, materialises its draw (one buffer, in the graph)
            from .bridge import rng_state
            self.rng = rng_state(0, dev)
            self.zc = z(1, 1) if corrector_name == "langevin" else None
            self.zp = None
        else:
            self.zc = z(self.N, max(self.n_steps, 1))
            self.zp = z(self.N)
        self.x_new, self.x_mean = z(), z()
        self.lstep = torch.zeros(2, B, device=dev)
        self.lscratch = torch.zeros(B * 128, dtype=torch.float64, device=dev)
        self.graph = None
        self.key = SamplerGraph._bridge_key(bridge)

    def _steps(self):
        prog, B = self.prog, self.B
        n = prog.x_in[0].numel()
        p = hip.ptr
        R = self.net.dense_rows
        self.dense_tab = prog.dense_table(self.t_tab, self.dense_bufs)      # (as in SamplerGraph._steps)
        draw = 1                       # (0 is the prior's)
        for i in range(self.N):
            # every evaluation at grid point i sees the same t: its Dense_0 rows go into place once
            hip.call("fdbm_copy_f32", p(prog.dense_out), p(self.dense_tab[i * B]), B * R)
            for k in range(self.n_steps):
                prog.run_body()
                c = self.ctab[i]
                step, nscale = c[3], c[4]
                zck = None if self.zc is None else (self.zc[0, 0] if self.rng is not None else self.zc[i, k])
                if self.corr == "langevin":
                    if self.rng is not None:
                        hip.call("fdbm_randn_complex", p(zck), B * n, p(self.rng), draw)
                    hip.call("fdbm_langevin_step", p(self.lstep[0]), p(self.lstep[1]), p(self.lscratch), p(prog.x_in),
                             p(prog.s_out), p(prog.y_in), p(zck), p(c[0]), p(c[1]), p(c[2]), self.snr, B, n)
                    step, nscale = self.lstep[0], self.lstep[1]
                if self.rng is not None and zck is None:
                    hip.call("fdbm_pc_corrector_rng", p(self.x_new), p(self.x_mean), p(prog.x_in), p(prog.s_out), p(prog.y_in),
                             p(self.rng), draw, p(c[0]), p(c[1]), p(c[2]), p(step), p(nscale), B, n)
                else:
                    hip.call("fdbm_pc_corrector", p(self.x_new), p(self.x_mean), p(prog.x_in), p(prog.s_out), p(prog.y_in),
                             p(zck), p(c[0]), p(c[1]), p(c[2]), p(step), p(nscale), B, n)
                draw += 1
                hip.call("fdbm_copy_f32", p(prog.x_in), p(self.x_new), 2 * B * n)
            if self.pred == "euler_maruyama":
                prog.run_body()
                w = self.ptab[i]
                if self.rng is not None:
                    hip.call("fdbm_pc_predictor_rng", p(self.x_new), p(self.x_mean), p(prog.x_in), p(prog.s_out), p(prog.y_in),
                             p(self.rng), draw, p(w[0]), p(w[1]), p(w[2]), p(w[3]), self.dts[i], B, n)
                else:
                    hip.call("fdbm_pc_predictor", p(self.x_new), p(self.x_mean), p(prog.x_in), p(prog.s_out), p(prog.y_in),
                             p(self.zp[i]), p(w[0]), p(w[1]), p(w[2]), p(w[3]), self.dts[i], B, n)
                draw += 1
                hip.call("fdbm_copy_f32", p(prog.x_in), p(self.x_new), 2 * B * n)
            else:
                hip.call("fdbm_copy_f32", p(self.x_mean), p(prog.x_in), 2 * B * n)      # NonePredictor returns (x, x): x_mean = x

    capture = SamplerGraph.capture

    def run(self, y, x0, noise, denoise):
        prog = self.prog
        # the reference draws: per grid point the corrector's noise (after its model call), then the predictor's (before
        # its model call) - the generator only sees the ORDER of the draws
        if self.rng is not None:
            from .bridge import rng_state
            self.rng.copy_(rng_state(noise.device_seed, "cpu"), non_blocking=True)
        else:
            for i in range(self.N):
                for k in range(self.n_steps):
                    self.zc[i, k].copy_(noise.step())
                if self.pred == "euler_maruyama":
                    self.zp[i].copy_(noise.step())
        prog.y_in.copy_(y)
        prog.x_in.copy_(x0)
        if self.graph is None:
            self.capture()
            prog.y_in.copy_(y)
            prog.x_in.copy_(x0)
        self.graph.replay()
        return (self.x_mean if denoise else prog.x_in).clone()


def pc_with_graph(net, bridge, y, noise, predictor_name, corrector_name, snr, n_steps, denoise):
    """Fast path of Bridge.pc_sampler for this package's backbone and the registered predictors / correctors."""
    B, _, F, T = y.shape
    key = ("pc", B, F, T, predictor_name, corrector_name, float(snr), int(n_steps), SamplerGraph._bridge_key(bridge), noise.in_kernel)
    pg = net._graphs.get(key)
    if pg is None:
        pg = net._graphs[key] = PcGraph(net, bridge, B, F, T, predictor_name, corrector_name, snr, n_steps,
                                        in_kernel_noise=noise.in_kernel)
    with torch.no_grad():
        x0 = bridge.prior_sampling(y.contiguous(), noise)
        return pg.run(y.contiguous(), x0, noise, denoise)


def sample_with_graph(net, bridge, y, kind, noise):
    """Fast path of Bridge.ode_sampler_ei / sde_sampler_ei for this package's backbone."""
    B, _, F, T = y.shape
    in_kernel = kind == "sde" and noise.in_kernel
    key = (kind, B, F, T, SamplerGraph._bridge_key(bridge), in_kernel)
    sg = net._graphs.get(key)
    if sg is None:
        sg = net._graphs[key] = SamplerGraph(net, bridge, kind, B, F, T, in_kernel_noise=in_kernel)
    with torch.no_grad():
        x0 = bridge.prior_sampling(y.contiguous(), noise)
        return sg.run(y.contiguous(), x0, noise.step if kind == "sde" else None, seed=noise.device_seed)
