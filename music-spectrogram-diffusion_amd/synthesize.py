"""Command line: MIDI file -> mel spectrogram of the whole song (the reference's notebook flow:
note_seq.midi_file_to_note_sequence -> full-song task pipeline -> InferSong.process,
beam/evaluation.py:161-223) on the MI355X path.

  python -m msd_amd.synthesize song.mid --checkpoint /path/to/base_with_context/checkpoint_500000 \\
      --out song_mel.npy [--preset base_with_context] [--gin-file train.gin --gin-bindings ...]
      [--seed 0] [--rng jax] [--num-steps 1000] [--dry-run]

--dry-run tokenises only (no GPU): prints the segment / token statistics the synthesis would see."""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def main(argv=None) -> int:
  ap = argparse.ArgumentParser(prog='msd_amd.synthesize', description=__doc__.split('\n')[0])
  ap.add_argument('midi')
  ap.add_argument('--checkpoint', default='synthetic:0',
                  help="T5X checkpoint dir, .npz/.safetensors flat dict, or 'synthetic[:seed]' (random weights)")
  ap.add_argument('--preset', default='base_with_context')
  ap.add_argument('--gin-file', default=None, help='training gin file (instead of --preset)')
  ap.add_argument('--gin-bindings', nargs='*', default=())
  ap.add_argument('--num-steps', type=int, default=1000)
  ap.add_argument('--cfg-weight', type=float, default=5.0)
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--rng', choices=['philox', 'jax'], default='philox')
  ap.add_argument('--precision', choices=['f16x3', 'f16', 'bf16x3', 'bf16'], default='f16x3',
                  help="'f16x3' (default): hi + lo IEEE-half operand planes, float32-class; 'bf16x3': bfloat16 planes "
                       "(float32's exponent range, twice the rounding error); 'f16' / 'bf16': one plane, not parity-grade")
  ap.add_argument('--no-range-fallback', dest='range_fallback', action='store_false',
                  help="fail with native.RangeError when an activation leaves the half-plane range (|x| > 65504) instead "
                       "of switching to 'bf16x3' and repeating the segment (the default: a real checkpoint's residual "
                       "stream may have outlier channels; the reference is float32 and never fails on them)")
  ap.add_argument('--out', default=None, help='.npy file for the mel frames [frames, 128]')
  ap.add_argument('--on-too-long', choices=['error', 'truncate'], default='error')
  ap.add_argument('--dry-run', action='store_true')
  args = ap.parse_args(argv)

  import msd_amd
  from msd_amd.frontend import midi_io, tokenizer
  if args.gin_file:
    spec = msd_amd.parse_training_gin_file(args.gin_file, list(args.gin_bindings))
  else:
    spec = msd_amd.config.preset(args.preset, num_steps=args.num_steps, cfg_weight=args.cfg_weight)
  ns = midi_io.midi_file_to_note_sequence(args.midi)
  cfg = tokenizer.FrontendConfig.from_spec(spec)
  t0 = time.perf_counter()
  segments = tokenizer.note_sequence_to_model_inputs(ns, cfg, on_too_long=args.on_too_long)
  t_tok = time.perf_counter() - t0
  n_tok = [int((s > 0).sum()) for s in segments]
  print('%s: %d notes, %.2f s -> %d segments of %d frames; tokens per segment min/mean/max %d/%.0f/%d (%.3f s)'
        % (args.midi, len(ns.notes), ns.total_time, len(segments), cfg.segment_frames, min(n_tok),
           float(np.mean(n_tok)), max(n_tok), t_tok), file=sys.stderr)
  if args.dry_run:
    return 0
  model = msd_amd.InferenceModel(args.checkpoint, spec, precision=args.precision, range_fallback=args.range_fallback)
  mel, timing = model.predict_sequence(segments, seed=args.seed, rng=args.rng, return_timing=True)
  frames = int(np.ceil(ns.total_time * cfg.frame_rate))
  mel = mel[0, :max(frames, 1)]
  print('synthesised %d mel frames; %.3f s per %.2f s segment (x%.2f realtime)'
        % (mel.shape[0], timing['prediction_seconds_per_chunk'], cfg.segment_frames / cfg.frame_rate,
           1.0 / timing['predictions_seconds_per_audio_second'] if timing['predictions_seconds_per_audio_second'] == timing['predictions_seconds_per_audio_second'] else float('nan')),
        file=sys.stderr)
  if args.out:
    np.save(args.out, mel)
  return 0


if __name__ == '__main__':
  sys.exit(main())
