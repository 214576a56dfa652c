"""Trial bookkeeping and EER (mirror of ``sidekit.bosaris`` for the scoring path)."""
from .idmap import IdMap
from .ndx import Ndx
from .key import Key
from .scores import Scores
from .detplot import effective_prior, logit_effective_prior, fast_minDCF, rocch, rocch2eer, pavx, rocch_from_histograms, eer_from_histograms
