"""ctypes binding of libsfmi.so (include/sfmi.h).

Fails loudly: if the HIP extension is missing or no GPU is usable there is NO
fallback -- importing works (so host-only helpers and symbol checks run on a
CPU box), but creating a batch raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SFMI_LIB_PATH: load a differently-built libsfmi (profiling / ablation builds); still HIP-only
LIB_PATH = os.environ.get("SFMI_LIB_PATH") or os.path.join(HERE, "libsfmi.so")

SF_OK = 0
SF_ERR_PRESET, SF_ERR_ARG, SF_ERR_HIP, SF_ERR_NO_DEVICE, SF_ERR_ACTION, SF_ERR_FIELD, SF_ERR_STATE = -1, -2, -3, -4, -5, -6, -7
OBS_TYPES = {"features": 0, "normalized-features": 1, "monitors": 2, "none": 3, "image": 4, "image-raw": 5}
IMAGE_W, IMAGE_H, IMAGE_OUT = 90, 92, 84
# SF_EV_* (include/sfmi.h): bit -> the reference's event string, in the order Game::stepOneTick can emit them
EVENT_NAMES = {0x1: "press-fire", 0x2: "press-thrust", 0x4: "press-left", 0x8: "press-right", 0x10: "release-fire",
               0x20: "release-thrust", 0x40: "release-left", 0x80: "release-right", 0x100: "missile-fired",
               0x200: "ship-respawn", 0x400: "explode-bighex", 0x800: "explode-smallhex", 0x1000: "fortress-respawn",
               0x2000: "fortress-fired", 0x4000: "shell-hit-ship", 0x8000: "hit-fortress", 0x10000: "vlner-increased",
               0x20000: "fortress-destroyed", 0x40000: "vlner-reset", 0x80000: "hit-dead-fortress",
               0x100000: "missile-left", 0x200000: "game-over"}
FLAG_OBS_F64 = 1
FLAG_REAL_SHELL_COUNT = 2
FLAG_NO_AUTO_RESET = 4
FLAG_REF_RESET_OBS = 8
EPISODE_STATS_LEN = 8


class SfmiError(RuntimeError):
    pass


class CreateParams(C.Structure):
    _fields_ = [
        ("gametype", C.c_char_p), ("n_envs", C.c_int32), ("device_id", C.c_int32), ("action_set", C.c_int32),
        ("obs_type", C.c_int32), ("flags", C.c_uint32), ("seed", C.c_uint32), ("spawn_skip", C.c_int32),
        ("spawn_stride", C.c_int32), ("spawn_table_len", C.c_int32),
    ]


class FieldDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("elem_size", C.c_int32), ("count", C.c_int32), ("is_float", C.c_int32)]


class Preset(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("game_time", C.c_int32),
        ("destroy_fortress", C.c_int32), ("ship_death_penalty", C.c_int32),
        ("missile_penalty", C.c_double), ("miss_penalty", C.c_int32),
        ("shell_speed", C.c_int32), ("shell_radius", C.c_int32), ("missile_speed", C.c_int32),
        ("missile_radius", C.c_int32), ("auto_turn", C.c_int32),
        ("sector_size", C.c_int32), ("lock_time", C.c_int32), ("vuln_time", C.c_int32),
        ("vuln_threshold", C.c_int32), ("fortress_radius", C.c_int32),
        ("big_hex", C.c_int32), ("small_hex", C.c_int32), ("explode_duration", C.c_int32),
        ("start_vx", C.c_double), ("start_vy", C.c_double), ("ship_radius", C.c_int32),
        ("ship_accel", C.c_double), ("turn_speed", C.c_int32), ("shaped", C.c_int32), ("n_keys", C.c_int32),
    ]


class ScoreGlyphs(C.Structure):
    """sf_score_glyphs (include/sfmi.h): the layout of a score-text glyph atlas"""
    _fields_ = [("gw", C.c_int32), ("gh", C.c_int32), ("advance", C.c_int32), ("y0", C.c_int32), ("x0", (C.c_int16 * 10) * 11)]


# every symbol include/sfmi.h declares: (restype, argtypes)
SYMBOLS = {
    "sf_create": (C.c_int, [C.POINTER(CreateParams), C.POINTER(C.c_void_p)]),
    "sf_destroy": (C.c_int, [C.c_void_p]),
    "sf_n_envs": (C.c_int, [C.c_void_p]),
    "sf_obs_dim": (C.c_int, [C.c_void_p]),
    "sf_n_actions": (C.c_int, [C.c_void_p]),
    "sf_tick_ms": (C.c_int, [C.c_void_p]),
    "sf_max_ticks": (C.c_int, [C.c_void_p]),
    "sf_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_check_actions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sf_check_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sf_clear_state_errors": (C.c_int, [C.c_void_p]),
    "sf_seed_actions": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "sf_step_sampled": (C.c_int, [C.c_void_p] * 7),
    "sf_rollout_sampled": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "sf_n_fields": (C.c_int, []),
    "sf_field_info": (C.c_int, [C.c_int, C.POINTER(FieldDesc)]),
    "sf_field_id": (C.c_int, [C.c_char_p]),
    "sf_get_field": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "sf_set_field": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "sf_get_field_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sf_episode_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sf_calibration_copy": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "sf_draw_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "sf_set_image_geometry": (C.c_int, [C.c_void_p] + [C.c_double] * 6),
    "sf_image_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sf_image_geometry_is_default": (C.c_int, [C.c_void_p]),
    "sf_set_score_glyphs": (C.c_int, [C.c_void_p, C.POINTER(ScoreGlyphs), C.c_void_p]),
    "sf_get_score_glyphs": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(ScoreGlyphs), C.c_void_p, C.c_size_t]),
    "sf_default_score_glyphs": (C.c_int, [C.POINTER(ScoreGlyphs), C.c_void_p, C.c_size_t]),
    "sf_preset_get": (C.c_int, [C.c_char_p, C.POINTER(Preset)]),
    "sf_action_table": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p]),
    "sf_spawn_table": (C.c_int, [C.c_uint32, C.c_int, C.c_void_p]),
    "sf_trig_table": (C.c_int, [C.c_void_p]),
    "sf_hex_points": (C.c_int, [C.c_int, C.c_void_p]),
    "sf_normalizer_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "sf_normalizer_destroy": (C.c_int, [C.c_void_p]),
    "sf_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sf_step_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "sf_normalizer_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_normalizer_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_step_record": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10),
    "sf_record_step_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p]),
    "sf_record_step": (C.c_int, [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p]),
    "sf_compute_returns": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "sf_render_stack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sf_render_shift": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "sf_frame_stack_clear": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "sf_set_event_output": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sf_render": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sf_image_background": (C.c_int, [C.c_void_p]),
    "sf_image_background_geom": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "sf_image_fort_alpha": (C.c_int, [C.c_int, C.c_void_p]),
    "sf_arc_table": (C.c_int, [C.c_void_p]),
    "sf_image_object_alpha": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_void_p]),
    "sf_image_explosion_host": (C.c_int, [C.c_double, C.c_double, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_void_p]),
    "sf_trig_deg": (C.c_int, [C.c_int, C.c_void_p]),
    "sf_image_arc_alpha": (C.c_int, [C.c_double] * 5 + [C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_void_p]),
    "sf_image_static": (C.c_int, [C.c_int, C.c_void_p]),
    "sf_resize_area_tab": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sf_resize_area_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "sf_set_render_order_hint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "sf_last_error": (C.c_char_p, []),
    "sf_version": (C.c_int, []),
    "sf_build_id": (C.c_char_p, []),
}

_lib = None


def lib():
    """Load libsfmi.so; raises SfmiError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfmiError(
                "libsfmi.so is missing (%s): build it with `python -m spacefortress_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if not hasattr(L, name) and os.environ.get("SFMI_LIB_PATH"):
                continue  # an older diagnostic build (A/B against an earlier round's library): entry points added since
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().sf_last_error().decode("utf-8", "replace")


def check(rc):
    """Map sf_status to the exception class the reference raises for the same misuse."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == SF_ERR_PRESET:
        raise RuntimeError(msg)  # SRC/pymodule.cpp:341
    if rc == SF_ERR_ARG:
        raise ValueError(msg)
    if rc == SF_ERR_ACTION:
        raise IndexError(msg)  # ENV:211-212
    if rc == SF_ERR_FIELD:
        raise KeyError(msg)
    if rc == SF_ERR_STATE:
        raise OverflowError(msg)
    raise SfmiError("libsfmi: %s (status %d)" % (msg, rc))


def raw_stream(device):
    """The current HIP stream of `device` as a ctypes pointer.  torch.cuda.current_stream(...).cuda_stream builds a
    Stream object per call (4 us of the 8 us a step launch costs on the host); the raw getter does not."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if get is not None:
        return C.c_void_p(get(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
