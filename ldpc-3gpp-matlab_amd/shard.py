"""Multi-GPU sharding of codeword batches: independent units, contiguous slices, no collective.

Codewords (and the C code blocks of a transport block, NRLDPCDecoder.m:257) share nothing but the
read-only shift tables, so rank g of `world` decodes codewords [lo, hi) of the batch on its own GPU
with its own handle.  torch.distributed is only used by callers for barriers / result gathering.

Two ways to drive several GPUs of one node:
  * one process per GPU (bench.py under torch.distributed.run): every rank builds its own Codec on its own device
    and decodes its shard_range(); no data-path collective, RCCL carries only the barrier and the max-over-ranks time;
  * one process, N GPUs (a MEX gateway inside one MATLAB process): _capi.CodecPool = nrldpc_pool_* in the C ABI: N
    handles, N host threads, 2-4 chunks per GPU pulled from a queue, so that GPUs whose codewords stop early take
    more work (BASELINE.json configs[4]).
"""


def shard_range(batch, rank, world):
    """Contiguous slice [lo, hi) of `batch` units for `rank`; sizes differ by at most one and the
    slices of ranks 0..world-1 tile [0, batch) in order."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("rank %d not in world of size %d" % (rank, world))
    base, extra = divmod(int(batch), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def decode_sharded(decode_fn, llr, rank, world):
    """Run decode_fn on this rank's slice of llr ([batch][n]); returns (lo, hi, result)."""
    lo, hi = shard_range(len(llr), rank, world)
    return lo, hi, decode_fn(llr[lo:hi])
