"""Whole-sampler HIP graphs: N x { backbone program ; fused state update } recorded once per
(shape, sampler, N, path) and replayed per batch of utterances.

The reference's inner loop (fdbm/bridge.py:73-85, :96-111) launches ~1.1k framework ops
plus ~45 tiny scalar kernels per step from Python.  Here one replay of one graph runs
all N steps: per step a 4-byte-per-sample copy of the model time, the recorded backbone
program (program.py) and ONE fdbm_bridge_update kernel that updates the state in place.
The per-step weights come from the host table (bridge.ei_weight_table) uploaded once.
"""
import torch

from . import hip


class SamplerGraph:
    def __init__(self, net, bridge, kind, B, F, T):
        self.net, self.kind, self.N = net, kind, bridge.N
        self.prog = net.program(B, F, T)
        dev = net.device
        table, t_model = bridge.ei_weight_table("ode" if kind == "ode" else "sde", B)
        self.table = table.to(dev).contiguous()                       # [N,3,B]
        # model time of each step as log t, evaluated on the host (see fdbm_temb)
        self.t_tab = torch.log(t_model[:, None] * torch.ones(1, B)).to(dev).contiguous()   # [N,B]
        self.z = None
        if kind == "sde":
            self.z = torch.zeros(self.N, B, 1, F, T, dtype=torch.complex64, device=dev)
        self.graph = None
        self.key = self._bridge_key(bridge)

    @staticmethod
    def _bridge_key(bridge):
        p = bridge.path
        return (type(p).__name__, bridge.N, bridge.start_time, bridge.end_time,
                tuple(sorted((k, v) for k, v in vars(p).items() if isinstance(v, (int, float, str)))))

    def _steps(self):
        prog = self.prog
        n = prog.x_in[0].numel()
        B = prog.B
        for i in range(self.N):
            hip.call("fdbm_copy_f32", hip.ptr(prog.t_in), hip.ptr(self.t_tab[i]), B)     # (a library kernel inside the graph)
            prog.run()
            third = prog.y_in if self.kind == "ode" else self.z[i]
            w = self.table[i]
            hip.call("fdbm_bridge_update", hip.ptr(prog.x_in), hip.ptr(prog.x_in), hip.ptr(prog.s_out),
                     hip.ptr(third), hip.ptr(w[0]), hip.ptr(w[1]), hip.ptr(w[2]), B, n)

    def capture(self):
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

    def run(self, y, x0, step_noise=None):
        """y, x0: complex64 [B,1,F,T] device tensors; returns the final state (new tensor)."""
        prog = self.prog
        prog.y_in.copy_(y)
        prog.x_in.copy_(x0)
        if self.kind == "sde":
            for i in range(self.N):
                self.z[i].copy_(step_noise())
        if self.graph is None:
            self.capture()
            prog.y_in.copy_(y)
            prog.x_in.copy_(x0)
        self.graph.replay()
        return prog.x_in.clone()


def sample_with_graph(net, bridge, y, kind, noise):
    """Fast path of Bridge.ode_sampler_ei / sde_sampler_ei for this package's backbone."""
    B, _, F, T = y.shape
    key = (kind, B, F, T, SamplerGraph._bridge_key(bridge))
    sg = net._graphs.get(key)
    if sg is None:
        sg = net._graphs[key] = SamplerGraph(net, bridge, kind, B, F, T)
    with torch.no_grad():
        x0 = bridge.prior_sampling(y.contiguous(), noise)
        return sg.run(y.contiguous(), x0, noise.step if kind == "sde" else None)
