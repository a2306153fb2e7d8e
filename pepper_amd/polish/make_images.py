"""`make_images(bam, fasta, region, output_dir, threads)`: unlabelled polish images
(/root/reference/pepper/modules/python/make_images.py:8-46)."""
import os

from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport


def make_images(bam_filepath, fasta_filepath, region, output_dir, threads, device_ids=None, stats=None, fused=None):
    """The reference's five arguments; device_ids ("0,1", a list, default device 0): worker t of the image generation uses
    device_ids[t % n]; stats: a dict the workers add their stage times to; fused: polish()'s fused consensus (polish/fused.py)."""
    if not os.path.isfile(bam_filepath):
        raise FileNotFoundError("CAN NOT LOCATE BAM FILE: " + str(bam_filepath))
    if not os.path.isfile(fasta_filepath):
        raise FileNotFoundError("CAN NOT LOCATE FASTA FILE: " + str(fasta_filepath))
    output_dir = UserInterfaceSupport.handle_output_directory(os.path.abspath(output_dir))
    if threads <= 0:
        raise ValueError("THREAD NEEDS TO BE >=0.")
    contig_list = UserInterfaceSupport.get_chromosome_list(region, fasta_filepath, bam_filepath, region_bed=None)
    UserInterfaceSupport.chromosome_level_parallelization(contig_list, bam_filepath, fasta_filepath, truth_bam=None,
                                                          output_path=output_dir, total_threads=threads, train_mode=False,
                                                          device_ids=device_ids, stats=stats, fused=fused)
