#!/usr/bin/env python3
"""Command line of the MI355X-native MicrobeCensus hot path; same flags, defaults and report file as the
reference's scripts/run_microbe_census.py (flags :14-55, flow :58-67), plus -g/--device."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microbecensus_amd  # noqa: E402
microbecensus_amd.configure_process_env()     # this program owns its process: eight HIP hardware queues, before anything touches the GPU
from microbecensus_amd import microbe_census  # noqa: E402


def parse_arguments(argv=None):
    p = argparse.ArgumentParser(usage="%s [-options] <seqfiles> <outfile>" % os.path.basename(__file__),
                                description="Estimate average genome size from metagenomic data (GPU search path).")
    p.add_argument("seqfiles", type=str, help="path to input metagenome(s); comma separated; FASTA/FASTQ, optionally gz/bz2")
    p.add_argument("outfile", type=str, help="path to the output report")
    p.add_argument("-v", dest="verbose", action="store_true", default=False, help="print program's progress to stdout")
    p.add_argument("-r", dest="rapsearch", type=str, default=None, help="path to an external RAPsearch2 v2.15 compatible executable to run instead of the in-process GPU search (the reference's hook)")
    p.add_argument("-n", dest="nreads", type=int, default=2000000, help="number of reads to sample (default = 2000000)")
    p.add_argument("-t", dest="threads", type=int, default=None, help="cap on the host threads of the read sampler (default: the machine's cores, up to 32; the reference's -t is the rapsearch thread count)")
    p.add_argument("-e", dest="no_equivs", action="store_true", default=False, help="skip the genome-equivalents pass over the input")
    p.add_argument("-l", dest="read_length", type=int, choices=microbe_census.VALID_READ_LENGTHS, help="trim all reads to this length")
    p.add_argument("-q", dest="min_quality", type=int, default=-5, help="minimum base-level PHRED quality (default = -5; no filtering)")
    p.add_argument("-m", dest="mean_quality", type=int, default=-5, help="minimum read-level PHRED quality (default = -5; no filtering)")
    p.add_argument("-d", dest="filter_dups", action="store_true", default=False, help="filter duplicate reads")
    p.add_argument("-u", dest="max_unknown", type=int, default=100, help="max percent of unknown bases per read (default = 100)")
    p.add_argument("-g", dest="device", type=int, default=None, help="GPU index (default: every visible GPU the run has batches of 2 M reads for)")
    args = vars(p.parse_args(argv))
    args["seqfiles"] = args["seqfiles"].split(",")
    if args["device"] is None:
        del args["device"]
    if args["threads"] is None:
        del args["threads"]                      # impute_missing_args() fills in the reference's default (1) for the report; the sampler is not capped
    return args


def main_distributed(args):
    """Launched by torchrun (one process per GPU): reads sharded over the ranks, one RCCL all_reduce of the per-family sums,
    rank 0 writes the report."""
    import torch
    import torch.distributed as dist
    from microbecensus_amd import distributed
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    try:
        est_ags, args = distributed.run_pipeline_distributed(args, device=local)
        if dist.get_rank() == 0:
            count_bases = microbe_census.count_bases(args) if not args["no_equivs"] else None
            microbe_census.report_results(args, est_ags, count_bases)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    args = parse_arguments()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:            # python -m torch.distributed.run --nproc-per-node N scripts/run_microbe_census.py ...
        main_distributed(args)
    else:
        est_ags, args = microbe_census.run_pipeline(args)
        count_bases = microbe_census.count_bases(args) if not args["no_equivs"] else None
        microbe_census.report_results(args, est_ags, count_bases)
