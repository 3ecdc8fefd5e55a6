"""MI355X-native watermark embed + detect hot path (drop-in for py/main16.py's Generator / Detector /
delta post-processing / loss stack).  Import name: ``awm_amd`` (see awm_amd.py at the repo root)."""
from .modules import Detector, Generator, ResBlock, load_state_dict_strip_prefix
from .losses import (MultiScaleMelLoss, TFLoudnessLoss, clamp_peak, detection_losses, fir_lowpass, high_freq_penalty,
                     l1_to_zero, limit_rms, postprocess)
from .step import LOSS_WEIGHTS, forward_losses, train_step, eval_forward
from .optim import FlatAdam
from .inference import (compute_si_snr, detect_prob, detect_watermark, detect_waveform, embed_waveform, evaluate_batches,
                        evaluate_unseen_file, generate_watermarked_audio, load_audio, lowpass_biquad, pcm16, save_audio, save_audio_float)
from . import checkpoint
from . import main14b_2
from . import distributed
from ._lib import lib, LIB_PATH

__all__ = ["Generator", "Detector", "ResBlock", "load_state_dict_strip_prefix", "MultiScaleMelLoss", "TFLoudnessLoss",
           "fir_lowpass", "clamp_peak", "limit_rms", "postprocess", "high_freq_penalty", "detection_losses", "l1_to_zero",
           "forward_losses", "train_step", "eval_forward", "LOSS_WEIGHTS", "FlatAdam", "distributed", "checkpoint", "main14b_2", "generate_watermarked_audio", "detect_watermark", "embed_waveform",
           "detect_waveform", "detect_prob", "evaluate_unseen_file", "evaluate_batches", "compute_si_snr", "load_audio", "save_audio", "save_audio_float", "lowpass_biquad", "pcm16", "lib", "LIB_PATH"]
