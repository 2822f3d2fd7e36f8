"""Architecture description of the NCSN++ v2 backbone family.

Produces the flat, ordered module list of the reference network
(fdbm/backbones/ncsnpp_v2.py:95-239, the ``all_modules`` ModuleList) as plain
data: one ``Mod`` per entry with its kind, channel counts and the names/shapes of
its parameters under the reference's state-dict keys
(``all_modules.<i>.<Sub>.<weight|bias|W|b>``, ``output_layer.*``).  Nothing here
touches a device; ``backbone.py`` compiles this list into a launch program.

Fixed options of every registered size (ncsnpp_v2.py:54-67): swish, BigGAN
res-blocks, fir=True with [1,3,3,1], skip_rescale, progressive output_skip /
input_skip with 'sum', fourier embedding scale 16, dropout 0.
"""
from dataclasses import dataclass, field

# name -> constructor kwargs (ncsnpp_v2.py:36-453)
VARIANTS = {
    "ncsnpp_v2": dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,)),
    "ncsnpp_v2_5M": dict(nf=96, ch_mult=(1, 1, 1, 1), num_res_blocks=1, attn_resolutions=(0,)),
    "ncsnpp_v2_16M": dict(nf=64, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(0,)),
    "ncsnpp_v2_37M": dict(nf=96, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,)),
}
IN_CH = 4      # x.re, x.im, y.re, y.im
OUT_CH = 2     # s.re, s.im
IMAGE_SIZE = 256


@dataclass
class Mod:
    idx: int                      # index in all_modules (-1: output_layer)
    kind: str                     # fourier|linear|conv3x3|resblock|attn|combine|groupnorm|conv1x1
    in_ch: int = 0
    out_ch: int = 0
    up: bool = False
    down: bool = False
    params: dict = field(default_factory=dict)   # state-dict key -> shape

    @property
    def prefix(self):
        return "output_layer" if self.idx < 0 else f"all_modules.{self.idx}"


def _gn(prefix, c):
    return {f"{prefix}.weight": (c,), f"{prefix}.bias": (c,)}


def _conv(prefix, cin, cout, k):
    return {f"{prefix}.weight": (cout, cin, k, k), f"{prefix}.bias": (cout,)}


def _lin(prefix, cin, cout):
    return {f"{prefix}.weight": (cout, cin), f"{prefix}.bias": (cout,)}


def _nin(prefix, cin, cout):
    return {f"{prefix}.W": (cin, cout), f"{prefix}.b": (cout,)}


def gn_groups(c):
    return min(c // 4, 32)


class Spec:
    """Ordered module list + derived bookkeeping for one backbone size."""

    def __init__(self, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                 attn_resolutions=(16,), **unused):
        self.nf = nf
        self.ch_mult = tuple(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.attn_resolutions = tuple(attn_resolutions)
        self.num_resolutions = len(self.ch_mult)
        self.all_resolutions = [IMAGE_SIZE // (2 ** i) for i in range(self.num_resolutions)]
        self.temb_dim = 4 * nf
        self.mods = []
        self._build()

    # ---- construction (mirrors ncsnpp_v2.py:95-239) ----------------------
    def _add(self, kind, **kw):
        m = Mod(idx=len(self.mods), kind=kind, **kw)
        self.mods.append(m)
        return m

    def _resblock(self, in_ch, out_ch=None, up=False, down=False):
        out_ch = out_ch or in_ch
        m = self._add("resblock", in_ch=in_ch, out_ch=out_ch, up=up, down=down)
        p = m.prefix
        m.params.update(_gn(f"{p}.GroupNorm_0", in_ch))
        m.params.update(_conv(f"{p}.Conv_0", in_ch, out_ch, 3))
        m.params.update(_lin(f"{p}.Dense_0", self.temb_dim, out_ch))
        m.params.update(_gn(f"{p}.GroupNorm_1", out_ch))
        m.params.update(_conv(f"{p}.Conv_1", out_ch, out_ch, 3))
        if in_ch != out_ch or up or down:
            m.params.update(_conv(f"{p}.Conv_2", in_ch, out_ch, 1))
        return m

    def _attn(self, c):
        m = self._add("attn", in_ch=c, out_ch=c)
        p = m.prefix
        m.params.update(_gn(f"{p}.GroupNorm_0", c))
        for j in range(4):
            m.params.update(_nin(f"{p}.NIN_{j}", c, c))
        return m

    def _build(self):
        nf = self.nf
        m = self._add("fourier", out_ch=2 * nf)
        m.params[f"{m.prefix}.W"] = (nf,)
        for cin in (2 * nf, 4 * nf):
            m = self._add("linear", in_ch=cin, out_ch=4 * nf)
            m.params.update(_lin(m.prefix, cin, 4 * nf))
        m = self._add("conv3x3", in_ch=IN_CH, out_ch=nf)
        m.params.update(_conv(m.prefix, IN_CH, nf, 3))

        hs_c = [nf]
        in_ch = nf
        for lvl in range(self.num_resolutions):
            for _ in range(self.num_res_blocks):
                out_ch = nf * self.ch_mult[lvl]
                self._resblock(in_ch, out_ch)
                in_ch = out_ch
                if self.all_resolutions[lvl] in self.attn_resolutions:
                    self._attn(in_ch)
                hs_c.append(in_ch)
            if lvl != self.num_resolutions - 1:
                self._resblock(in_ch, down=True)
                m = self._add("combine", in_ch=IN_CH, out_ch=in_ch)
                m.params.update(_conv(f"{m.prefix}.Conv_0", IN_CH, in_ch, 1))
                hs_c.append(in_ch)

        in_ch = hs_c[-1]
        self._resblock(in_ch)
        self._attn(in_ch)
        self._resblock(in_ch)

        for lvl in reversed(range(self.num_resolutions)):
            for _ in range(self.num_res_blocks + 1):
                out_ch = nf * self.ch_mult[lvl]
                self._resblock(in_ch + hs_c.pop(), out_ch)
                in_ch = out_ch
            if self.all_resolutions[lvl] in self.attn_resolutions:
                self._attn(in_ch)
            m = self._add("groupnorm", in_ch=in_ch, out_ch=in_ch)
            m.params.update(_gn(m.prefix, in_ch))
            m = self._add("conv3x3", in_ch=in_ch, out_ch=IN_CH)
            m.params.update(_conv(m.prefix, in_ch, IN_CH, 3))
            if lvl != 0:
                self._resblock(in_ch, up=True)
        assert not hs_c
        self.output_layer = Mod(idx=-1, kind="conv1x1", in_ch=IN_CH, out_ch=OUT_CH)
        self.output_layer.params.update(_conv("output_layer", IN_CH, OUT_CH, 1))

    # ---- bookkeeping -------------------------------------------------------
    def param_shapes(self):
        """state-dict key -> shape, in module order (the reference's key set)."""
        out = {}
        for m in self.mods:
            out.update(m.params)
        out.update(self.output_layer.params)
        return out

    def param_order(self):
        """Keys in the order of the reference module's `parameters()` (= its state_dict order; it has no buffers):
        `output_layer` is registered before `all_modules` (ncsnpp_v2.py:93,239).  This is the order of torch_ema's
        `shadow_params` in a Lightning checkpoint; pinned by tests/golden/param_order.json."""
        return list(self.output_layer.params) + [k for m in self.mods for k in m.params]

    def num_params(self):
        n = 0
        for shp in self.param_shapes().values():
            k = 1
            for d in shp:
                k *= d
            n += k
        return n

    def macs_per_forward(self, F=256, T=256):
        """Algorithmic multiply-accumulates of one forward for one sample at F x T
        (conv + NIN + attention + linear; elementwise excluded) - the figure
        SURVEY.md 8(d) quotes: 266 073 636 864 for ncsnpp_v2 at 256 x 256."""
        from .program import count_macs
        return count_macs(self, F, T)
