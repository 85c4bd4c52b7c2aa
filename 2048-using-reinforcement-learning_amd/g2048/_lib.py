"""ctypes binding of the C-ABI in include/g2048.h (csrc/libg2048_hip.so).

There is no fallback: if the library is missing or no HIP device is visible, every
compute call raises. PyTorch is only used for device memory and streams -- tensors
are passed as data_ptr(), work is enqueued on torch's current stream.
"""
import ctypes as C
import os

import torch

from . import _build

_lib = None

OK = 0
ABI_VERSION = 5
FLAG_DONE, FLAG_VALID, FLAG_MAXCODE_SHIFT = 0x01, 0x02, 3
STEP_REWARD_F64, STEP_AUTO_RESET, STEP_RANDOM_ACTIONS, STEP_NOOP_ACTIONS = 0x01, 0x02, 0x04, 0x08
VALID_ENV, VALID_AGENT = 0, 1
(EVAL_FAST, EVAL_FULL, EVAL_PPO_HEURISTIC, EVAL_MONO_PP, EVAL_MONO_PM, EVAL_MONO_MP, EVAL_MONO_MM, EVAL_PPO_SHAPING, EVAL_PATTERN,
 EVAL_CORNER_BONUS, EVAL_MERGE_POTENTIAL) = range(11)
BEAM_FIXED_DOWN = 0x01
BEAM_RANK_BY_COUNTING = 0x04
PLAY_ONE_PHASE = 0x02
BEAM_MAX_WIDTH = 128
KEYBLOCK_WORDS = 16
ROLLOUT_OBS_SHIFT, OBS_F32, OBS_F16, OBS_BF16 = 4, 0, 1, 2
SEEN_SLOT_BYTES = 32
ENV_RECORD_BYTES, ENV_OP_STEP, ENV_OP_RESET, ENV_OP_PEEK, ENV_OP_MOVE, ENV_OP_SPAWN, ENV_OP_MOVE_AGENT = 80, 0, 1, 2, 3, 4, 5
ENV_TOKEN_SHIFT = 8

_vp, _u64, _sz, _u32, _int = C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint32, C.c_int
SIGNATURES = {
    "g2048_last_error": (C.c_char_p, []),
    "g2048_abi_version": (_int, []),
    "g2048_build_flags": (C.c_uint, []),
    "g2048_device_count": (_int, []),
    "g2048_step": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _sz, _u32, _vp]),
    "g2048_step_many": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _u32, _u64, _sz, _u32, _vp]),
    "g2048_reset": (_int, [_vp, _vp, _u64, _u64, _u64, _sz, _vp]),
    "g2048_valid_moves": (_int, [_vp, _vp, _sz, _u32, _vp]),
    "g2048_eval": (_int, [_vp, _int, _vp, _vp, _sz, _vp]),
    "g2048_obs_f32": (_int, [_vp, _vp, _sz, _vp]),
    "g2048_obs_16": (_int, [_vp, _vp, _int, _sz, _vp]),
    "g2048_beam_get_action": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _u64, _u64, _u64, _sz, _u32, _vp]),
    "g2048_beam_workspace_bytes": (_sz, [_sz]),
    "g2048_beam_get_action_ws": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _u64, _u64, _u64, _sz, _u32, _vp, _sz, _vp]),
    "g2048_beam_history_bytes": (_sz, [_sz]),
    "g2048_beam_get_action_hist": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _u64, _u64, _u64, _sz, _u32, _vp, _sz, _u32, _vp]),
    "g2048_track_episodes": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _sz, _vp]),
    "g2048_sample_actions": (_int, [_vp, _vp, _vp, _vp, _u64, _u64, _u64, _sz, _vp]),
    "g2048_simulate_move": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "g2048_simulate_move_sampled": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _sz, _vp]),
    "g2048_play_games": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _u64, _u64, _sz, _u32, _vp]),
    "g2048_play_games_workspace": (_sz, [_sz]),
    "g2048_play_games_ws": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _u64, _u64, _sz, _u32, _vp,
                                  _sz, _vp]),
    "g2048_play_games_tuned": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _u64, _u64, _sz, _u32,
                                     _vp, _sz, _vp, _vp]),
    "g2048_replay_games": (_int, [_vp, _vp, _vp, _u64, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _u64, _sz, _vp]),
    "g2048_env_step": (_int, [_vp, _vp, _u32, _u32, _vp, _u64, _u64, _u64, _vp]),
    "g2048_minibatch_gather": (_int, [_vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _sz, _sz, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp]),
    "g2048_launch_plan": (_int, [_int, _int, _sz, _vp]),
    "g2048_device_plan": (_int, [_int, _sz, _vp]),
    "g2048_pack_i32": (_int, [_vp, _vp, _sz, _vp]),
    "g2048_unpack_i32": (_int, [_vp, _vp, _sz, _vp]),
    "g2048_synth_boards": (_int, [_vp, _u64, _u64, _sz, _u32, _u32, _vp]),
    "g2048_synth_actions": (_int, [_vp, _u64, _u64, _u64, _sz, _vp]),
    "g2048_metrics": (_int, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "g2048_keys_advance": (_int, [_vp, _vp, _u64, _vp]),
    "g2048_step_dyn": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _sz, _u32, _vp]),
    "g2048_beam_get_action_dyn": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _u64, _sz, _u32, _vp]),
    "g2048_sample_actions_dyn": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _sz, _vp]),
    "g2048_track_episodes_dyn": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "g2048_selftest": (_int, [_vp, _vp]),
    "g2048_sort_selftest": (_int, [_vp, _vp, _sz, _int, _vp]),
    "g2048_rollout_step": (_int, [_vp] * 13 + [_u64, _u64, _vp, _u64, _sz, _u32, _vp]),
    "g2048_shaping_scan_workspace": (_sz, [_sz]),
    "g2048_shaping_scan": (_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "g2048_seen_insert": (_int, [_vp, _u64, _vp, _u32, _vp, _vp, _vp, _sz, _vp]),
    "g2048_seen_rehash": (_int, [_vp, _u32, _vp, _u32, _vp, _vp]),
    "g2048_shaping_apply": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _sz, _vp]),
}


def library_path():
    return _build.LIB


def lib():
    """The loaded C-ABI library. Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("g2048: %s is missing -- build it with `python __graft_entry__.py` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        if os.environ.get("G2048_LIB"):      # A/B builds (tools/build_ab.sh) are only ever loaded on request, and say so
            import sys
            print("g2048: loading the A/B build %s (G2048_LIB is set)" % path, file=sys.stderr)
        elif os.path.basename(path) != "libg2048_hip.so":
            raise RuntimeError("g2048: refusing to load %s: only csrc/libg2048_hip.so is the product library" % path)
        L = C.CDLL(path)        # torch is imported above, so libamdhip64 resolves to the runtime torch uses
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError here = ABI mismatch, deliberately loud
            fn.restype, fn.argtypes = res, args
        if L.g2048_abi_version() != ABI_VERSION:
            raise RuntimeError("g2048: ABI version mismatch (library %d, binding %d)" % (L.g2048_abi_version(), ABI_VERSION))
        flags = int(L.g2048_build_flags())
        if flags and os.environ.get("G2048_ALLOW_INSTRUMENTED") != "1":
            # a measurement build (csrc/g2048_instrument.h) overwrites real outputs with clock ticks: only the timeline tools,
            # which set G2048_ALLOW_INSTRUMENTED=1 themselves, may load one
            raise RuntimeError("g2048: %s is an instrumented measurement build (g2048_build_flags() = 0x%x); its outputs are "
                               "not results -- refusing to load it (tools/*_timeline.py opt in with G2048_ALLOW_INSTRUMENTED=1)"
                               % (path, flags))
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise RuntimeError("g2048: %s (status %d)" % (lib().g2048_last_error().decode(), rc))


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def call(device, fn, *args):
    """Invoke a C-ABI entry point with `device` as the calling thread's current HIP device (a launch goes to the
    current device, which need not be the tensors' device in a one-process multi-GPU program) and check its status."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if torch.cuda.current_device() == idx:
        rc = fn(*args)
    else:
        with torch.cuda.device(idx):
            rc = fn(*args)
    check(rc)


def require_device_tensor(t, dtype, shape_tail=None, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError("g2048: %s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("g2048: %s must live on a ROCm device (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("g2048: %s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("g2048: %s must be contiguous" % name)
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError("g2048: %s must have shape (n,%s) (got %s)" % (name, ",".join(map(str, shape_tail)), tuple(t.shape)))
    return t


def u64(x):
    return int(x) & 0xFFFFFFFFFFFFFFFF
