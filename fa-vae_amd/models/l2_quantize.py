"""Cosine-similarity vector quantiser on the MI355X-native kernels.

Drop-in for the reference's `models/l2_quantize.py` on the path every shipped script uses (`--use_l2_quantizer
--use_cosine_sim`, heads=1, no k-means init, no dead-code expiry, no orthogonal regulariser):
  VectorQuantize.forward      l2_quantize.py:533-596      CosineSimCodebook.forward   l2_quantize.py:391-444
  get_codebook_entry          l2_quantize.py:518-530      buffers/keys                l2_quantize.py:343-350
The (tokens x codes) similarity matrix and the one-hot matrix of the reference are never materialised: the arg-max
is fused into the tiled MFMA product, the EMA sums are a deterministic segmented scatter (csrc/vq.hip).
Options of the upstream lucidrains class that no FA-VAE script enables raise NotImplementedError instead of
silently running something else.
"""
import torch
import torch.distributed as distributed
import torch.nn.functional as F
from torch import nn

from favae_hip import ops as K


def l2norm(t):
    return F.normalize(t, p=2, dim=-1)


def uniform_init(*shape):
    t = torch.empty(shape)
    nn.init.kaiming_uniform_(t)
    return t


class CosineSimCodebook(nn.Module):
    def __init__(self, dim, codebook_size, num_codebooks=1, kmeans_init=False, kmeans_iters=10, decay=0.8, eps=1e-5,
                 threshold_ema_dead_code=2, use_ddp=False, learnable_codebook=False, sample_codebook_temp=0.0):
        super().__init__()
        if num_codebooks != 1 or kmeans_init or learnable_codebook or sample_codebook_temp != 0.0:
            raise NotImplementedError("only the single, EMA-updated, arg-max codebook of the FA-VAE scripts is accelerated")
        if threshold_ema_dead_code != 0:
            raise NotImplementedError("dead-code expiry is never enabled by FA-VAE (threshold_ema_dead_code=0)")
        self.decay = decay
        self.codebook_size = codebook_size
        self.num_codebooks = num_codebooks
        self.eps = eps
        self.use_ddp = use_ddp
        self.tie_eps = 4e-6          # top-2 gaps below this are re-scored in fp64 (SURVEY 7 "bit-exact indices")
        self.register_buffer("initted", torch.Tensor([True]))
        self.register_buffer("cluster_size", torch.zeros(num_codebooks, codebook_size))
        self.register_buffer("embed", l2norm(uniform_init(num_codebooks, codebook_size, dim)))

    # comm_timing: when a list, every codebook all-reduce appends (bytes, start event, end event) -- bench.py's `comm` object and
    # tests/dist_probe.py read them after a synchronize (what the first real multi-GPU run has to show: models/l2_quantize.py:419,427)
    comm_timing = None

    def _all_reduce(self, t):
        if self.use_ddp:
            rec = self.comm_timing
            if rec is not None and t.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                distributed.all_reduce(t)
                e1.record()
                rec.append((t.numel() * t.element_size(), e0, e1))
            else:
                distributed.all_reduce(t)

    @torch.no_grad()
    def forward(self, x):
        """x: (b, n, d) tokens -> (quantize (b, n, d) raw codebook rows, embed_ind (b, n) int64)."""
        b, n, d = x.shape
        tokens = x.float().reshape(b * n, d).contiguous()
        embed = self.embed[0]
        idx, zq, zn, en = K.vq_lookup(tokens, embed, self.tie_eps)
        if self.training:
            bins, esum = K.vq_segment_sum(zn, idx, self.codebook_size)
            self._all_reduce(bins)                          # l2_quantize.py:419
            self._all_reduce(esum)                          # l2_quantize.py:427
            K.vq_ema_update(embed, self.cluster_size[0], en, bins, esum, self.decay)
        return zq.view(b, n, d), idx.view(b, n)


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size, codebook_dim=None, heads=1, separate_codebook_per_head=False, decay=0.8, eps=1e-5,
                 kmeans_init=False, kmeans_iters=10, use_cosine_sim=False, threshold_ema_dead_code=0, channel_last=True,
                 accept_image_fmap=False, commitment_weight=1.0, orthogonal_reg_weight=0.0,
                 orthogonal_reg_active_codes_only=False, orthogonal_reg_max_codes=None, sample_codebook_temp=0.0,
                 sync_codebook=False):
        super().__init__()
        if not use_cosine_sim:
            raise NotImplementedError("EuclideanCodebook is not on the FA-VAE hot path (all scripts pass --use_cosine_sim)")
        if heads != 1 or orthogonal_reg_weight > 0:
            raise NotImplementedError("multi-head / orthogonal-regularised codebooks are never enabled by FA-VAE")
        if not accept_image_fmap:
            raise NotImplementedError("VQGANFCM always passes accept_image_fmap=True")
        self.heads = heads
        self.separate_codebook_per_head = separate_codebook_per_head
        codebook_dim = codebook_dim if codebook_dim is not None else dim
        requires_projection = codebook_dim != dim
        self.project_in = nn.Linear(dim, codebook_dim) if requires_projection else nn.Identity()
        self.project_out = nn.Linear(codebook_dim, dim) if requires_projection else nn.Identity()
        self.eps = eps
        self.commitment_weight = commitment_weight
        self.orthogonal_reg_weight = orthogonal_reg_weight
        self._codebook = CosineSimCodebook(dim=codebook_dim, num_codebooks=1, codebook_size=codebook_size, kmeans_init=kmeans_init,
                                           kmeans_iters=kmeans_iters, decay=decay, eps=eps,
                                           threshold_ema_dead_code=threshold_ema_dead_code, use_ddp=sync_codebook,
                                           learnable_codebook=False, sample_codebook_temp=sample_codebook_temp)
        self.codebook_size = codebook_size
        self.accept_image_fmap = accept_image_fmap
        self.channel_last = channel_last
        self._cfg1 = K.ConvCfg(1, 1, 1, 0)

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def get_codebook_entry(self, indices, shape):
        z_q = self._codebook.embed[0][indices.reshape(-1)]          # == one-hot @ embed (l2_quantize.py:520-523)
        if shape is not None:
            z_q = z_q.view(shape).permute(0, 3, 1, 2).contiguous()
        return z_q

    def forward(self, x):
        """x (B, C, H, W) -> quantize (B, C, H, W), embed_ind (B, H, W) int64, loss (1,)."""
        B, C, H, W = x.shape
        if isinstance(self.project_in, nn.Linear):
            x = K.fused_conv(x, self.project_in.weight, self.project_in.bias, cfg=self._cfg1)
        x = K.as_cl(x)
        d = x.shape[1]
        tokens = x.permute(0, 2, 3, 1).reshape(B, H * W, d)          # 'b c h w -> b (h w) c' is a free view of NHWC memory
        zq, idx = self._codebook(tokens.detach())
        if self.training:
            quantize, loss = K.VQStraightThroughFn.apply(tokens, zq, float(self.commitment_weight))
        else:
            quantize, loss = zq, torch.zeros(1, device=x.device)
        quantize = quantize.view(B, H, W, d).permute(0, 3, 1, 2)      # back to (B, d, H, W), channels-last memory
        if isinstance(self.project_out, nn.Linear):
            quantize = K.fused_conv(quantize, self.project_out.weight, self.project_out.bias, cfg=self._cfg1)
        return quantize, idx.view(B, H, W), loss
