"""wav.scp -> x-vector ark/scp: the driver of ``sidekit/bin/extract_xvectors.py`` on the MI355X path.

Same arguments and files (``--model --wav-scp --out-scp [--out-spk-scp --spk2utt-file] --device
--sample-rate``).  Differences, all on the caller's side of ``Xtractor.forward``: utterances are
length-sorted into padded batches (``--batch-size``, each row is still processed as if run alone),
decoding, host -> device copies and the forward overlap (``sidekit_amd/pipeline.py``); under
``python -m torch.distributed.run --nproc-per-node N -m sidekit_amd.bin.extract_xvectors ...`` the wav.scp is sharded over N GPUs
(one all-gather of the x-vectors, rank 0 writes),
the bf16 trunk can be selected (``--dtype bf16``), ``--vad`` is refused (the reference fetches Silero
VAD with ``torch.hub`` at run time, ``extract_xvectors.py:102`` -- no network here).  PCM wavs are
decoded with ``scipy.io.wavfile`` (``soundfile`` is not installed); ``cmd |`` entries are run through
the shell exactly as the reference does (``:57-70``).
"""
import argparse
import io
import os
import subprocess

import numpy
import scipy.io.wavfile
import torch
import torch.distributed as dist

from ..kaldi_io import ArkScpWriter, OrderedArkWriter, read_scp
from ..nnet.xvector import Xtractor
from ..pipeline import StreamingExtractor, host_workers
from ..sharding import gather_xvectors, shard_range


def read_wav_scp(wav_scp):
    utt2wav = {}
    with open(wav_scp) as f:
        for line in f:
            parts = line.strip().split()
            if parts:
                utt2wav[parts[0]] = parts[1:]
    return utt2wav


def prepare(wav):
    """One wav.scp entry -> (float32 1-D tensor in [-1, 1), sample rate)."""
    wav = ' '.join(wav)
    if wav.strip().endswith("|"):
        try:
            out = subprocess.run(wav.strip()[:-1], shell=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            sr, sample = scipy.io.wavfile.read(io.BytesIO(out))
        except Exception as e:
            raise IOError("Error processing wav file: {}\n{}".format(wav, e))
    else:
        sr, sample = scipy.io.wavfile.read(wav)
    if sample.ndim > 1:
        raise IOError(f"{wav}: expected a mono file, got shape {sample.shape}")
    if sample.dtype == numpy.int16:
        sample = sample.astype(numpy.float32) / 32768.0
    elif sample.dtype == numpy.int32:
        sample = sample.astype(numpy.float32) / 2147483648.0
    elif sample.dtype == numpy.uint8:
        sample = (sample.astype(numpy.float32) - 128.0) / 128.0
    return torch.from_numpy(numpy.ascontiguousarray(sample, dtype=numpy.float32)), sr


def load_model(model_path, device):
    """``extract_xvectors.py:74-89``: checkpoint dict -> Xtractor on ``device`` in eval mode."""
    device = torch.device(device)
    checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
    archi = checkpoint["model_archi"]
    emb = archi["embedding_size"] if "embedding_size" in archi else 256
    xtractor = Xtractor(checkpoint["speaker_number"], model_archi=archi["model_type"], loss=archi["loss"]["type"], embedding_size=emb)
    xtractor.load_state_dict(checkpoint["model_state_dict"], strict=True)
    return xtractor.to(device).eval(), archi


def precheck(xtractor, entries, sample_rate, workers=4):
    """Header-only pass over the plain files of ``entries`` (``cmd |`` pipes cannot be probed): raises ``IOError`` for a file that
    cannot be opened and ``ValueError`` for one too short for the front-end's reflect padding (``n_fft / 2 < samples``, what
    ``torch.stft(center=True)`` demands in the reference) at the model's rate."""
    from ..pipeline import probe_wavs
    plain = [(k, src.strip()) for k, src in entries if not src.strip().endswith("|")]
    if not plain:
        return
    kind, ns, rate, _ = probe_wavs([p for _, p in plain], workers)
    pre = getattr(xtractor, "preprocessor", None)
    need = getattr(pre, "n_fft", 0) // 2
    for (key, path), k, n, r in zip(plain, kind, ns, rate):
        if k < 0:
            raise IOError(f"Error processing wav file: {path} ({key}): cannot be opened")
        if k == 1 and need and -(-int(n) * sample_rate // max(int(r), 1)) <= need:
            raise ValueError(f"{key}: {int(n)} samples at {int(r)} Hz is too short for the front-end (needs more than {need} samples at {sample_rate} Hz)")


@torch.no_grad()
def main(xtractor, wav_scp, out_file, device, sample_rate=16000, out_file_spk="", spk2utt_file="", batch_size=64, dtype="fp32",
         workers=None, window=8, gather_always=False):
    """One process: the whole wav.scp.  Under ``torch.distributed.run`` (one process per GPU, an initialised process group):
    every rank streams the contiguous shard ``shard_range(len(wav.scp), rank, world)``, the ``(N_r, E)`` blocks are gathered once
    (``gather_xvectors``: RCCL all-gather, ragged counts) and rank 0 writes the ark / scp files in wav.scp order.
    With ONE rank the files are written incrementally as in a plain run (``gather_always`` takes the collective path all the same: the
    rehearsal of the N-rank code on a one-GPU box).  ``workers=None``: this rank's share of the host cores (``host_workers``)."""
    if workers is None:
        workers = host_workers()
    utt2wav = read_wav_scp(wav_scp)
    xtractor.compute_dtype = dtype
    keys = list(utt2wav)
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(len(keys), rank, world)
    # every plain file of the shard is probed (native header walk, milliseconds per thousand files) BEFORE any GPU work: an unreadable or
    # too-short file stops the run here, not after hours of extraction -- and under torch.distributed.run not while the other
    # ranks already sit in the all-gather
    precheck(xtractor, [(key, ' '.join(utt2wav[key])) for key in keys[lo:hi]], sample_rate, workers)
    # decode threads -> length-sorted batches inside a sliding window -> pinned staging -> copy stream -> forward, all at
    # once (sidekit_amd/pipeline.py); every row is still computed over its own length
    stream = StreamingExtractor(xtractor, batch_size=batch_size, window=window, workers=workers, sample_rate=sample_rate)
    results = stream.run((key, ' '.join(utt2wav[key])) for key in keys[lo:hi])
    out_ark = os.path.realpath(os.path.join(os.path.dirname(out_file), os.path.splitext(os.path.basename(out_file))[0]))
    sharded = dist.is_initialized() and (world > 1 or gather_always)
    if not sharded:
        # one process (or one rank): the ark holds its records in wav.scp order like the reference's (extract_xvectors.py:120,147) although
        # batches come back length-sorted -- a (1, E) record's size follows from its key, so every x-vector is written at its final offset
        # the moment its batch arrives (flushed per batch: what was extracted survives an interruption); the scp grows in arrival order
        # beside it and is rewritten in wav.scp order at the end
        with OrderedArkWriter(f"{out_ark}.ark", os.path.realpath(out_file), keys, xtractor.embedding_size) as writer:
            for n, (key, vec) in enumerate(results):
                writer(key, vec)
                if n % batch_size == batch_size - 1:
                    writer.flush()
        vecs = None
    mine = dict(results) if sharded else None
    if sharded:
        dev = torch.device(xtractor.device)
        block = numpy.concatenate([mine[k] for k in keys[lo:hi]]) if hi > lo else numpy.zeros((0, xtractor.embedding_size), dtype=numpy.float32)
        full = gather_xvectors(torch.from_numpy(numpy.ascontiguousarray(block, dtype=numpy.float32)).to(dev)).cpu().numpy()
        if rank != 0:
            return
        vecs = {k: full[i:i + 1] for i, k in enumerate(keys)}      # contiguous shards in rank order = wav.scp order
        with ArkScpWriter(f"{out_ark}.ark", os.path.realpath(out_file)) as writer:
            for key in utt2wav:              # wav.scp order, as the reference
                writer(key, vecs[key])
    if out_file_spk:                         # speaker means, L2-normalised (:153-173)
        spk2utt = {}
        with open(spk2utt_file) as f:
            for line in f:
                parts = line.strip().split()
                spk2utt[parts[0]] = parts[1:]
        utt2embd = dict(read_scp(out_file))
        out_ark_spk = os.path.realpath(os.path.join(os.path.dirname(out_file_spk), os.path.splitext(os.path.basename(out_file_spk))[0]))
        with ArkScpWriter(f"{out_ark_spk}.ark", os.path.realpath(out_file_spk)) as writer:
            for spk, utts in spk2utt.items():
                mean = numpy.mean([utt2embd[u] for u in utts], axis=0)
                mean /= numpy.linalg.norm(mean, ord=2)
                writer(spk, mean)


def cli(argv=None):
    parser = argparse.ArgumentParser(description="Extract the x-vectors given a sidekit model (MI355X path)")
    parser.add_argument("--model", type=str, required=True, help="SideKit model checkpoint")
    parser.add_argument("--sample-rate", type=int, default=16000, help="Must match SideKit SR model")
    parser.add_argument("--vad", action='store_true', help="not available: the reference downloads Silero VAD at run time")
    parser.add_argument("--wav-scp", type=str, required=True)
    parser.add_argument("--out-scp", type=str, required=True)
    parser.add_argument("--out-spk-scp", type=str, default="")
    parser.add_argument("--spk2utt-file", type=str, default="")
    parser.add_argument("--device", default="cuda", type=str)
    parser.add_argument("--batch-size", type=int, default=64)
    parser.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    parser.add_argument("--workers", type=int, default=0, help="wav decoding threads (0: this rank's share of the host cores, at most 8)")
    parser.add_argument("--gather-always", action="store_true", help="with one rank under torch.distributed.run: still gather through the collective and let rank 0 "
                        "write at the end (rehearsal of the N-rank path on a one-GPU box) instead of writing incrementally")
    parser.add_argument("--window", type=int, default=8, help="utterances are length-sorted inside windows of this many batches")
    args = parser.parse_args(argv)
    assert os.path.isfile(args.model), "NO SUCH FILE: %s" % args.model
    assert os.path.isfile(args.wav_scp), "NO SUCH FILE: %s" % args.wav_scp
    assert os.path.isdir(os.path.dirname(args.out_scp)), "NO SUCH DIRECTORY: %s" % args.out_scp
    if args.vad:
        raise NotImplementedError("--vad needs torch.hub.load('snakers4/silero-vad') (remote fetch): out of scope")
    if args.out_spk_scp:
        assert os.path.isdir(os.path.dirname(args.out_spk_scp)), "NO SUCH DIRECTORY: %s" % args.out_spk_scp
        assert os.path.isfile(args.spk2utt_file), "NO SUCH FILE: %s" % args.spk2utt_file
    device = args.device.strip().lower()
    # python -m torch.distributed.run --nproc-per-node N -m sidekit_amd.bin.extract_xvectors ...: the launcher's whole environment, not a
    # stray RANK a batch system exported (a rendezvous without MASTER_ADDR / MASTER_PORT fails or hangs)
    if all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if device.startswith("cuda"):
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(local)
            device = f"cuda:{local}"
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")
    xtractor, _ = load_model(args.model, device)
    main(xtractor, args.wav_scp, args.out_scp, device, args.sample_rate, args.out_spk_scp, args.spk2utt_file, args.batch_size, args.dtype, args.workers or None, args.window, args.gather_always)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    cli()
