"""StyleGAN2 ResNet discriminator on the vqk kernels -- the `resnet` / fp32-default configuration that
``VQLPIPSWithDiscriminator`` instantiates (``Discriminator(image_size)``, loss.py:69 of the reference).

Same class names, constructor arguments, parameter names/shapes and ``state_dict`` keys as the reference's
``stylegan2_discriminator/discriminator.py`` (FullyConnectedLayer :92-121, Conv2dLayer :127-174,
DiscriminatorBlock :180-265, MinibatchStdLayer :271-293, DiscriminatorEpilogue :299-354, Discriminator :360-412).
Where the reference chains ``conv2d_resample`` -> ``bias_act`` plugin calls, here one NHWC conv launch carries the
runtime weight gain, bias, leaky-ReLU and output gain in its epilogue; the FIR blur / decimation is the NHWC
``upfirdn2d`` kernel; the stride-2 3x3 conv runs as a true strided implicit GEMM.
Not built: 'orig' / 'skip' architectures, conditioning (c_dim > 0), conv_clamp, fp16 blocks (all unreachable from
loss.py)."""
import numpy as np
import torch
from torch import nn

from ... import ops
from ..autoencoder import _to_internal

_ACT_GAIN = {'linear': 1.0, 'lrelu': float(np.sqrt(2))}


def setup_filter(f):
    """normalised separable-as-outer-product FIR (upfirdn2d.py:72-116 for the 1-D < 8 taps case)"""
    f = torch.as_tensor(f, dtype=torch.float32)
    f = torch.outer(f, f)
    return f / f.sum()


class FullyConnectedLayer(nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier
        if self.bias_gain != 1:
            raise NotImplementedError('lr_multiplier != 1 is not on the discriminator path')

    def forward(self, x):
        """x [B, in] -> [B, out]: a 1x1 conv over B 'pixels'"""
        b, f = x.shape
        y = ops.conv_act(x.reshape(b, f, 1, 1), self.weight.reshape(self.weight.shape[0], f, 1, 1), self.bias, k=1,
                         act=self.activation, wgain=self.weight_gain, out_gain=_ACT_GAIN[self.activation])
        return y[:, :self.weight.shape[0], 0, 0]


class Conv2dLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        if up != 1 or conv_clamp is not None or not trainable or down not in (1, 2):
            raise NotImplementedError('only the discriminator configuration (up=1, down in {1,2}) is built')
        self.activation, self.up, self.down = activation, up, down
        self.kernel_size = kernel_size
        self.register_buffer('resample_filter', setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = _ACT_GAIN[activation]
        w = torch.randn([out_channels, in_channels, kernel_size, kernel_size])
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros([out_channels])) if bias else None

    def forward(self, x, gain=1):
        k, f = self.kernel_size, self.resample_filter
        kw = dict(k=k, act=self.activation, wgain=self.weight_gain, out_gain=self.act_gain * gain)
        if self.down == 1:
            return ops.conv_act(x, self.weight, self.bias, stride=1, pad=self.padding, **kw)
        fw = f.shape[0]
        p0 = self.padding + (fw - self.down + 1) // 2          # conv2d_resample.py:100-104
        p1 = self.padding + (fw - self.down) // 2
        if k == 1:                                             # :107-110  blur+decimate, then 1x1
            x = ops.upfirdn2d_nhwc(x, f, down=self.down, padding=(p0, p1, p0, p1))
            return ops.conv_act(x, self.weight, self.bias, stride=1, pad=0, **kw)
        x = ops.upfirdn2d_nhwc(x, f, padding=(p0, p1, p0, p1))  # :119-122  blur, then strided conv without padding
        return ops.conv_act(x, self.weight, self.bias, stride=self.down, pad=0, **kw)


class DiscriminatorBlock(nn.Module):
    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx,
                 architecture='resnet', activation='lrelu', resample_filter=(1, 3, 3, 1), conv_clamp=None,
                 use_fp16=False, fp16_channels_last=False, freeze_layers=0):
        super().__init__()
        if architecture != 'resnet' or use_fp16 or freeze_layers:
            raise NotImplementedError("only architecture='resnet' in full precision is built")
        assert in_channels in [0, tmp_channels]
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.first_layer_idx, self.architecture = first_layer_idx, architecture
        self.register_buffer('resample_filter', setup_filter(list(resample_filter)))
        self.num_layers = 0
        if in_channels == 0:
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation)
            self.num_layers += 1
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2,
                                 resample_filter=resample_filter)
        self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2,
                                resample_filter=resample_filter)
        self.num_layers += 3

    def forward(self, x, img, double_backward: bool = False):
        if self.in_channels == 0:
            x = self.fromrgb(img)
        c0, c1, sk = self.conv0, self.conv1, self.skip
        if (ops.FUSE_DISC_BLOCK and not double_backward and torch.is_grad_enabled() and c0.activation == 'lrelu'
                and c1.resample_filter.shape == (4, 4)):
            # one autograd node per block (ops.DiscBlockFn): same forward launches, fused backward passes
            fw = c1.resample_filter.shape[0]
            pb0, pb1 = c1.padding + (fw - 1) // 2, c1.padding + (fw - 2) // 2      # Conv2dLayer.forward's p0 / p1 with down = 2
            ps0, ps1 = sk.padding + (fw - 1) // 2, sk.padding + (fw - 2) // 2
            cfg = (float(c0.weight_gain), float(c1.weight_gain), float(sk.weight_gain), ops.ACT_CODE['lrelu'], float(c0.act_gain),
                   float(np.sqrt(0.5)), (pb0, pb1, pb0, pb1), (ps0, ps1, ps0, ps1))
            return ops.DiscBlockFn.apply(x, c0.weight, c0.bias, c1.weight, c1.bias, sk.weight,
                                         c1.resample_filter.to(torch.float32).contiguous(), cfg), None
        y = self.skip(x, gain=np.sqrt(0.5))
        x = self.conv0(x)
        x = self.conv1(x, gain=np.sqrt(0.5))
        return ops.AddFn.apply(y, x), None


class MinibatchStdLayer(nn.Module):
    def __init__(self, group_size, num_channels=1):
        super().__init__()
        if num_channels != 1:
            raise NotImplementedError('one minibatch-stddev feature (the default) is built')
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x, halves: int = 1):
        """halves > 1: x is `halves` independent batches concatenated (real | fake in one discriminator pass): the statistic
        groups samples inside each batch exactly as separate calls would (discriminator.py:277-293)"""
        if halves > 1:
            return torch.cat([ops.MbstdFn.apply(c, self.group_size if self.group_size is not None else c.shape[0])
                              for c in x.chunk(halves, 0)], 0)
        return ops.MbstdFn.apply(x, self.group_size if self.group_size is not None else x.shape[0])


class DiscriminatorEpilogue(nn.Module):
    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4,
                 mbstd_num_channels=1, activation='lrelu', conv_clamp=None):
        super().__init__()
        if architecture != 'resnet' or cmap_dim != 0:
            raise NotImplementedError("only architecture='resnet' without conditioning is built")
        self.in_channels, self.cmap_dim, self.resolution = in_channels, cmap_dim, resolution
        self.mbstd = MinibatchStdLayer(mbstd_group_size, mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1)

    def forward(self, x, img=None, cmap=None, halves: int = 1):
        if self.mbstd is not None:
            x = self.mbstd(x, halves)
        x = self.conv(x)[:, :self.in_channels]
        x = x.permute(0, 1, 2, 3).reshape(x.shape[0], -1)      # flatten(1) of the LOGICAL NCHW tensor (c, h, w order)
        x = self.fc(x)
        return self.out(x).to(torch.float32)


class Discriminator(nn.Module):
    def __init__(self, img_resolution, c_dim=0, img_channels=3, architecture='resnet', channel_base=32768, channel_max=512,
                 num_fp16_res=0, conv_clamp=None, cmap_dim=None, block_kwargs=None, mapping_kwargs=None,
                 epilogue_kwargs=None):
        super().__init__()
        if c_dim != 0 or num_fp16_res != 0:
            raise NotImplementedError('unconditional full-precision discriminator only')
        self.compute_dtype = torch.float32
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        ch = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        cur = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(ch[res] if res < img_resolution else 0, ch[res], ch[res // 2], resolution=res,
                                       first_layer_idx=cur, img_channels=img_channels, architecture=architecture,
                                       **(block_kwargs or {}))
            setattr(self, f'b{res}', block)
            cur += block.num_layers
        self.b4 = DiscriminatorEpilogue(ch[4], cmap_dim=0, resolution=4, img_channels=img_channels,
                                        architecture=architecture, **(epilogue_kwargs or {}))

    def forward(self, img, halves: int = 1, double_backward: bool = False, **_):
        """halves: number of independent batches concatenated along dim 0 (only the minibatch-stddev layer couples samples);
        double_backward: this pass will be differentiated twice (R1, loss.py:98-112): layer-by-layer autograd nodes instead of
        the fused block nodes, whose backward is first-order only"""
        ops.set_conv_products(getattr(self, 'conv_products', 'fp32'))
        img = _to_internal(img, self.compute_dtype)
        x = None
        for res in self.block_resolutions:
            x, _unused = getattr(self, f'b{res}')(x, img, double_backward)
        return self.b4(x, halves=halves)
