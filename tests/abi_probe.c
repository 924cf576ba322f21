/* Prints size and field offsets of `struct frame_headers`.  Compiled twice by
 * tests/test_abi.py: against include/mlvfs_abi.h, and (only where /root/reference
 * exists) with -DUSE_REFERENCE against the reference's own mlvfs.h. */
#include <stddef.h>
#include <stdio.h>
#ifdef USE_REFERENCE
#include "mlvfs.h"
#else
#include "mlvfs_abi.h"
#endif
#define P(f) printf(#f " %zu\n", offsetof(struct frame_headers, f))
int main(void)
{
    printf("sizeof_frame_headers %zu\nsizeof_raw_info %zu\n", sizeof(struct frame_headers), sizeof(struct raw_info));
    P(fileNumber); P(position); P(vidf_hdr); P(vidf_hdr.panPosX); P(vidf_hdr.panPosY); P(file_hdr);
    P(file_hdr.fileGuid); P(file_hdr.videoClass); P(rtci_hdr); P(idnt_hdr); P(idnt_hdr.cameraModel); P(rawi_hdr);
    P(rawi_hdr.xRes); P(rawi_hdr.yRes); P(rawi_hdr.raw_info); P(rawi_hdr.raw_info.height); P(rawi_hdr.raw_info.width);
    P(rawi_hdr.raw_info.frame_size); P(rawi_hdr.raw_info.bits_per_pixel); P(rawi_hdr.raw_info.black_level);
    P(rawi_hdr.raw_info.white_level); P(rawi_hdr.raw_info.active_area); P(rawi_hdr.raw_info.exposure_bias);
    P(rawi_hdr.raw_info.cfa_pattern); P(rawi_hdr.raw_info.color_matrix1); P(rawi_hdr.raw_info.dynamic_range);
    P(expo_hdr); P(lens_hdr); P(wbal_hdr);
    return 0;
}
