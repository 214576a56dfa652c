"""Neural extractor surface (mirror of ``sidekit.nnet`` for the hot path)."""
from .preprocessor import MelSpecFrontEnd, MfccFrontEnd
from .xvector import Xtractor, extract_embeddings, extract_embeddings_per_speaker, test_metrics
