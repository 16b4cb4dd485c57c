"""Loader of the native boundary: ``libvqk.so`` (C-ABI declared in ``include/vqk.h``).

Counterpart of the reference's plugin loader ``custom_ops.get_plugin``
(vqvae/modules/loss/stylegan2_discriminator/utils/custom_ops.py:49-129): there the two CUDA
plugins are JIT-built with ninja/nvcc and a *silent fallback* to a PyTorch implementation exists;
here the library is built ahead of time, in-tree, with plain ``hipcc --offload-arch=gfx950``
(one builder at a time, file lock), loaded with ``ctypes`` and there is NO fallback: a missing or
broken library raises.
"""
from __future__ import annotations

import ctypes
import fcntl
import os
import subprocess
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
_SO = os.environ.get('VQK_LIB') or os.path.join(_HERE, 'libvqk.so')     # VQK_LIB: A/B builds of the same ABI (tools/)
_lib = None

P = c_void_p
I, L, F = c_int, c_int64, c_float

# name -> argtypes ; every function returns int (status) unless listed in _SPECIAL
_PROTOS = {
    'vqk_row_sqnorm_f32': [P, L, I, P, P],
    'vqk_vq_assign_f32': [P, P, P, P, L, I, I, I, P, P],
    'vqk_vq_assign_filtered_f32': [P, P, P, P, L, I, I, I, P, P, L, P],
    'vqk_probe_stream_add': [P, P, L, I, I, I, P],
    'vqk_calib_fill': [P, L, P],
    'vqk_calib_mfma': [P, L, P, I, I, P],
    'vqk_calib_copy': [P, P, L, P],
    'vqk_vq_prepare_f32': [P, I, I, P, L, P],
    'vqk_vq_forward_f32': [P, P, P, L, L, I, I, I, P, P, P, P, P, P],
    'vqk_vq_backward_fused_f32': [P, P, P, P, I, L, I, I, F, F, P, P, P, P],
    'vqk_vq_distances_f32': [P, P, P, P, L, I, I, I, P, P, P],
    'vqk_vq_distances_stats_f32': [P, P, P, P, L, I, I, I, P, P, F, P, P, P, P],
    'vqk_entropy_forward_presummed_f32': [P, L, I, F, P, P, P, P, P],
    'vqk_entropy_forward_f32': [P, L, I, F, P, P, P, P, P, P, P],
    'vqk_entropy_backward_f32': [P, P, P, P, L, I, F, F, P, P],
    'vqk_entropy_backward_split_f32': [P, P, P, P, L, I, F, F, P, P, P, P],
    'vqk_entropy_argmax_forward_f32': [P, P, P, L, I, F, P, P, P, P, P, P, P, P],
    'vqk_entropy_argmax_backward_f32': [P, P, P, P, P, L, I, F, F, P, P],
    'vqk_row_scale_add_f32': [P, P, P, L, I, F, P],
    'vqk_gumbel_forward': [I, P, P, L, I, F, I, P, P, P, P, P, P],
    'vqk_gumbel_backward': [I, P, P, P, L, I, F, F, P, P, P, P],
    'vqk_vq_gather_f32': [P, P, P, L, I, I, P, P, P, P, P],
    'vqk_vq_backward_f32': [P, P, P, P, I, L, I, I, F, F, P, P, P, P],
    'vqk_ema_stats_f32': [P, P, L, I, I, P, P, P],
    'vqk_ema_stats_fused_f32': [P, P, L, I, I, P, P, P],
    'vqk_ema_update_f32': [P, P, P, P, P, I, I, F, F, F, P],
    'vqk_conv2d_fprop': [I, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, P],
    'vqk_conv2d_fprop_pooled': [I, P, P, P, P, P, I, I, I, I, I, I, I, F, P, P],
    'vqk_conv2d_fprop_gnstats': [I, P, P, P, P, P, I, I, I, I, I, I, I, I, F, P, I, P, P],
    'vqk_conv2d_ups_phase': [I, P, P, P, P, I, I, I, I, I, I, P, I, P, P],
    'vqk_conv2d_pooled_dgrad_phase': [I, P, P, P, I, I, I, I, I, F, P, P],
    'vqk_conv2d_pooled_fprop_phase': [I, P, P, P, P, I, I, I, I, I, F, P, I, P, P],
    'vqk_conv2d_thin_in_gnstats': [I, P, P, P, P, I, I, I, I, P, I, P],
    'vqk_conv2d_s2_supported': [I, I, I, I, I, I, I],
    'vqk_conv2d_s2_fprop': [I, P, P, P, P, I, I, I, I, I, I, F, F, P, P],
    'vqk_conv2d_s2_dgrad': [I, P, P, P, P, I, I, I, I, I, F, P, P],
    'vqk_conv2d_general': [I, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, F, F, I, P, P],
    'vqk_conv2d_wgrad_general': [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P],
    'vqk_conv2d_wgrad_general_scaled': [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, F, P, P],
    'vqk_conv_weight_layout': [I, I, I, I, I, I, I, I],
    'vqk_conv_pack_weights': [P, P, I, I, I, I, I, I, P],
    'vqk_conv_pack_multi': [P, I, I, P],
    'vqk_conv_set_variant': [I],
    'vqk_conv_set_block_caps': [I, I],
    'vqk_set_deterministic': [I, P, L],
    'vqk_set_scratch': [P, L],
    'vqk_set_tile_queue': [P, L],
    'vqk_ctx_create': [P],
    'vqk_ctx_destroy': [P],
    'vqk_ctx_make_current': [P],
    'vqk_ctx_set_scratch': [P, P, L],
    'vqk_ctx_set_tile_queue': [P, P, L],
    'vqk_ctx_set_deterministic': [P, I, P, L],
    'vqk_conv_pack_dgrad': [P, P, I, I, I, I, P],
    'vqk_conv2d_wgrad': [I, P, P, P, I, I, I, I, I, I, I, P, P],
    'vqk_conv2d_wgrad_pooled_dy': [I, P, P, P, I, I, I, I, I, F, P, P],
    'vqk_conv2d_wgrad_ups_phase': [I, P, P, P, I, I, I, I, I, F, P, P],
    'vqk_split_pair_f32': [P, P, L, I, P],
    'vqk_conv2d_fprop_x3_gnstats': [P, P, P, P, P, I, I, I, I, I, I, P, I, P, P],
    'vqk_conv2d_wgrad_x3': [P, P, P, I, I, I, I, I, I, F, P, P],
    'vqk_conv2d_wgrad_x3_f32': [P, P, P, I, I, I, I, I, I, F, P],
    'vqk_conv2d_wgrad_pooled_dy_phase': [I, P, P, P, I, I, I, I, I, F, P, P],
    'vqk_conv2d_wgrad_edge': [I, P, P, P, P, L, I, I, I, I, I, P, P],
    'vqk_conv2d_wgrad_edge_true': [I, P, P, P, P, L, I, I, I, I, I, I, P, P],
    'vqk_conv2d_thin_out': [I, P, P, P, P, I, I, I, I, I, I, P, P],
    'vqk_colsum': [I, P, L, I, P, P],
    'vqk_colsum_lead': [I, P, L, I, I, F, P, P],
    'vqk_cast': [P, P, I, L, P],
    'vqk_gn_stats': [I, P, I, L, I, I, F, P, P, P],
    'vqk_gn_apply': [I, P, P, P, P, P, I, L, I, I, I, P],
    'vqk_gn_forward': [I, P, P, P, P, P, P, I, L, I, I, F, I, P],
    'vqk_gn_forward_presummed': [I, P, P, P, P, P, P, I, L, I, I, F, I, P],
    'vqk_gn_forward_presummed_parts': [I, P, P, P, P, P, P, I, P, I, L, I, I, F, I, P],
    'vqk_gn_backward': [I, P, P, P, P, P, P, P, P, P, I, L, I, I, I, I, P, P],
    'vqk_gn_backward_pooled_add': [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P, F, P],
    'vqk_gn_backward_ws': [I, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, I, P, P, F, P],
    'vqk_gn_cluster_timeouts': [P],
    'vqk_gn_backward_colsum': [I, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, I, P, P, P],
    'vqk_pool2x2': [I, P, P, I, I, I, I, F, P],
    'vqk_unpool2x2': [I, P, P, I, I, I, I, F, P],
    'vqk_preprocess': [P, P, I, P, I, I, I, I, P],
    'vqk_augment_preprocess': [P, P, P, P, I, P, I, I, I, I, P],
    'vqk_sse': [I, P, P, L, P, P],
    'vqk_pair_stats': [P, P, L, P, P],
    'vqk_ssim_sum': [P, P, I, I, I, I, P, I, P, F, F, P, P],
    'vqk_mse_tanh_backward': [I, P, P, L, F, P, I, P, P],
    'vqk_tanh_backward': [I, P, P, P, L, P],
    'vqk_axpby': [I, P, P, P, F, F, L, P],
    'vqk_adamw': [P, P, P, P, L, P, P, I, F, F, F, F, I, F, P, P],
    'vqk_act_backward': [I, P, P, P, L, I, F, P],
    'vqk_act_backward_colsum': [I, P, P, P, L, I, I, F, P, P],
    'vqk_act_backward_colsum_scaled': [I, P, P, P, L, I, I, F, F, P, P],
    'vqk_upfirdn2d_nhwc': [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, F, I, I, P],
    'vqk_upfirdn2d_act_backward': [I, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, I, P],
    'vqk_maxpool2x2': [I, P, P, P, I, I, I, I, I, P],
    'vqk_channel_affine': [I, P, P, P, P, L, I, P],
    'vqk_lpips_tap': [I, P, P, P, I, L, I, P, P, F, P, P],
    'vqk_mbstd': [I, P, P, P, P, I, L, I, I, I, I, P],
    'vqk_mbstd_double_backward': [I, P, P, P, P, P, I, L, I, I, I, P],
    'vqk_l1_sum': [I, P, P, L, P, P],
    'vqk_l1l2_backward': [I, P, P, L, F, F, P, P, I, P],
    'vqk_gan_loss': [P, P, I, I, I, P, P, P, P, P],
    'vqk_bias_act': [P, P, P, P, P, P, L, L, I, I, I, F, F, F, P],
    'vqk_upfirdn2d': [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, F, I, I, P],
}
_SPECIAL = {'vqk_set_tuning': (I, [c_char_p, I]), 'vqk_reset_tuning': (I, []), 'vqk_tuning_count': (I, []),
            'vqk_tuning_name': (c_char_p, [I]),
            'vqk_conv_packed_elems': (c_int64, [I, I, I, I]), 'vqk_calib_mfma_flops': (c_int64, [I, I]), 'vqk_conv2d_wgrad_edge_ws_bytes': (c_int64, []), 'vqk_vq_filter_ws_bytes': (c_int64, [I, I]), 'vqk_status_str': (c_char_p, [I]), 'vqk_version': (I, []), 'vqk_arch': (c_char_p, [])}
EXPORTS = sorted(list(_PROTOS) + list(_SPECIAL))


def library_path() -> str:
    return _SO


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``libvqk.so`` (cross-compiles without a GPU).

    Serialised across processes with a file lock (the reference serialises its JIT build with a
    FileBaton, custom_ops.py:100-110)."""
    lock_path = os.path.join(_CSRC, '.build.lock')
    with open(lock_path, 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(('.hip', '.cpp', '.h', '.inc'))]
            srcs.append(os.path.join(_HERE, '..', 'include', 'vqk.h'))
            stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
            if force or stale:
                cmd = ['make', '-C', _CSRC, '-j8'] + ([] if verbose else ['-s'])
                if force:
                    subprocess.check_call(['make', '-C', _CSRC, '-s', 'clean'])
                subprocess.check_call(cmd)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


def lib() -> ctypes.CDLL:
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- loads the HIP runtime (libamdhip64.so.7) the library binds to
    if not os.path.exists(_SO):
        raise RuntimeError(f'vqk: native library {_SO} is missing -- run `python -c "import __graft_entry__ as g; '
                           f'g.build()"` (hipcc --offload-arch=gfx950). There is no fallback path.')
    cdll = ctypes.CDLL(_SO, mode=ctypes.RTLD_GLOBAL)
    for name, args in _PROTOS.items():
        fn = getattr(cdll, name)
        fn.restype = c_int
        fn.argtypes = args
    for name, (res, args) in _SPECIAL.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    _lib = cdll
    apply_env_tuning(cdll)
    return _lib


def apply_env_tuning(cdll=None, environ=None) -> dict:
    """The library reads no environment variable (include/vqk.h); its launch heuristics are tuning slots.  This host layer
    keeps the interface of the A/B scripts under tools/: every ``VQK_<SLOT>`` found in the environment is handed to
    ``vqk_set_tuning`` once, when the library is loaded (VQK_WGMX_COEF is a float, its slot counts 1e-4 units; the two
    formerly presence-only switches VQK_GN_NO_SMALL / VQK_WGRAD_NO_* count as 1 when set to anything but '0')."""
    cdll = cdll or lib()
    environ = os.environ if environ is None else environ
    applied = {}
    for i in range(cdll.vqk_tuning_count()):
        name = cdll.vqk_tuning_name(i).decode()
        env = 'VQK_WGMX_COEF' if name == 'WGMX_COEF_E4' else 'VQK_' + name
        raw = environ.get(env)
        if raw is None:
            continue
        if name == 'WGMX_COEF_E4':
            val = int(round(float(raw) * 1e4))
        elif name in ('GN_NO_SMALL', 'WGRAD_NO_PW16', 'WGRAD_NO_P16K'):
            val = 0 if raw == '0' else 1
        else:
            val = int(raw)
        check(cdll.vqk_set_tuning(name.encode(), val), f'set_tuning({name})')
        applied[name] = val
    return applied


HOST_SWITCHES: dict = {}        # name -> (default, value in effect): every VQK_* switch of the HOST layer (ops.py), read once at import


def switch(name: str, default: str) -> str:
    """A/B switch of the host layer (``VQK_<NAME>`` in the environment, else ``default``).  The operators read their switches
    through here -- one registry next to the library's tuning slots (``apply_env_tuning``), none scattered as ``os.environ`` reads
    -- and ``HOST_SWITCHES`` lists what a process runs with."""
    val = os.environ.get(name, default)
    HOST_SWITCHES[name] = (default, val)
    return val


ERR_SHAPE = -1          # VQK_ERR_SHAPE (include/vqk.h)


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f'vqk: {what} failed: {lib().vqk_status_str(status).decode()} ({status})')
