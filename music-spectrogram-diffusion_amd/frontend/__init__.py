"""MIDI / NoteSequence front end of the synthesis path: note events ->
int32 token segments in exactly the reference's vocabulary and segmentation, so that
InferenceModel.predict_sequence is MIDI-in.  Host-side Python like the reference's
(event_codec.py, vocabularies.py, note_sequences.py, run_length_encoding.py, the full-song chain of
preprocessors.py / tasks.py); no TensorFlow, seqio, t5 or note_seq needed.  Bit-exact integer work,
pinned by the reference's own golden vectors (tests/test_frontend_*.py)."""
from . import event_codec, vocabularies, note_sequences, run_length_encoding, midi_io, tokenizer  # noqa: F401
