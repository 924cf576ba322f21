/*
 * mlvfs_abi.h -- binary layout of the parameter block MLVFS passes to every
 * hot-path function: `struct frame_headers`.
 *
 * Replaces (layout-compatible, declared independently):
 *   mlvfs/mlvfs.h:51-63   struct frame_headers
 *   mlvfs/mlv.h:40-239    the MLV v2.0 block headers it embeds (#pragma pack(1))
 *   mlvfs/raw.h:166-207   struct raw_info
 * The MLV block layouts are the public Magic Lantern MLV v2.0 file format.  Only
 * the blocks that `frame_headers` embeds are declared.  tests/test_abi.py checks
 * every offset and size against the reference's own headers (when
 * /root/reference is present) and against the values frozen below.
 *
 * A maintainer building MLVFS against libmlvfs_amd.so keeps using MLVFS's own
 * headers; this file only has to agree with them byte for byte.
 */
#ifndef MLVFS_ABI_H
#define MLVFS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sensor/raw description delivered by Magic Lantern's raw backend */
struct raw_info {
    uint32_t api_version;
    uint32_t do_not_use_this;          /* 32-bit buffer pointer slot, unused on 64-bit hosts */
    int32_t  height, width, pitch;
    int32_t  frame_size;
    int32_t  bits_per_pixel;
    int32_t  black_level;
    int32_t  white_level;
    union {
        struct { int32_t x, y, width, height; } jpeg;
        struct { int32_t origin[2], size[2]; } crop;
    };
    union {
        struct { int32_t y1, x1, y2, x2; } active_area;
        int32_t dng_active_area[4];
    };
    int32_t  exposure_bias[2];
    int32_t  cfa_pattern;
    int32_t  calibration_illuminant1;
    int32_t  color_matrix1[18];
    int32_t  dynamic_range;
};

#pragma pack(push, 1)

/* every MLV block starts with: 4-byte tag, 32-bit size, 64-bit timestamp */
#define MLV_BLOCK_PREFIX uint8_t blockType[4]; uint32_t blockSize; uint64_t timestamp

typedef struct {                       /* "MLVI" file header */
    uint8_t  fileMagic[4];
    uint32_t blockSize;
    uint8_t  versionString[8];
    uint64_t fileGuid;
    uint16_t fileNum, fileCount;
    uint32_t fileFlags;
    uint16_t videoClass, audioClass;
    uint32_t videoFrameCount, audioFrameCount;
    uint32_t sourceFpsNom, sourceFpsDenom;
} mlv_file_hdr_t;

typedef struct {                       /* "VIDF" video frame */
    MLV_BLOCK_PREFIX;
    uint32_t frameNumber;
    uint16_t cropPosX, cropPosY;
    uint16_t panPosX, panPosY;
    uint32_t frameSpace;
} mlv_vidf_hdr_t;

typedef struct {                       /* "RAWI" raw format */
    MLV_BLOCK_PREFIX;
    uint16_t xRes, yRes;
    struct raw_info raw_info;
} mlv_rawi_hdr_t;

typedef struct {                       /* "EXPO" exposure */
    MLV_BLOCK_PREFIX;
    uint32_t isoMode, isoValue, isoAnalog, digitalGain;
    uint64_t shutterValue;
} mlv_expo_hdr_t;

typedef struct {                       /* "LENS" */
    MLV_BLOCK_PREFIX;
    uint16_t focalLength, focalDist, aperture;
    uint8_t  stabilizerMode, autofocusMode;
    uint32_t flags, lensID;
    uint8_t  lensName[32], lensSerial[32];
} mlv_lens_hdr_t;

typedef struct {                       /* "RTCI" wall-clock time */
    MLV_BLOCK_PREFIX;
    uint16_t tm_sec, tm_min, tm_hour, tm_mday, tm_mon, tm_year, tm_wday, tm_yday, tm_isdst, tm_gmtoff;
    uint8_t  tm_zone[8];
} mlv_rtci_hdr_t;

typedef struct {                       /* "IDNT" camera identity */
    MLV_BLOCK_PREFIX;
    uint8_t  cameraName[32];
    uint32_t cameraModel;
    uint8_t  cameraSerial[32];
} mlv_idnt_hdr_t;

typedef struct {                       /* "WBAL" white balance */
    MLV_BLOCK_PREFIX;
    uint32_t wb_mode, kelvin, wbgain_r, wbgain_g, wbgain_b, wbs_gm, wbs_ba;
} mlv_wbal_hdr_t;

#pragma pack(pop)

/* all block headers that belong to one video frame (natural alignment) */
struct frame_headers {
    uint32_t        fileNumber;
    uint64_t        position;
    mlv_vidf_hdr_t  vidf_hdr;
    mlv_file_hdr_t  file_hdr;
    mlv_rtci_hdr_t  rtci_hdr;
    mlv_idnt_hdr_t  idnt_hdr;
    mlv_rawi_hdr_t  rawi_hdr;
    mlv_expo_hdr_t  expo_hdr;
    mlv_lens_hdr_t  lens_hdr;
    mlv_wbal_hdr_t  wbal_hdr;
};

#ifdef __cplusplus
}
#define MLVFS_ABI_ASSERT(c, m) static_assert(c, m)
#else
#define MLVFS_ABI_ASSERT(c, m) _Static_assert(c, m)
#endif

/* frozen sizes (checked against the reference headers by tests/test_abi.py) */
MLVFS_ABI_ASSERT(sizeof(struct raw_info) == 160, "raw_info");
MLVFS_ABI_ASSERT(sizeof(mlv_file_hdr_t) == 52, "MLVI");
MLVFS_ABI_ASSERT(sizeof(mlv_vidf_hdr_t) == 32, "VIDF");
MLVFS_ABI_ASSERT(sizeof(mlv_rawi_hdr_t) == 180, "RAWI");
MLVFS_ABI_ASSERT(sizeof(mlv_expo_hdr_t) == 40, "EXPO");
MLVFS_ABI_ASSERT(sizeof(mlv_lens_hdr_t) == 96, "LENS");
MLVFS_ABI_ASSERT(sizeof(mlv_rtci_hdr_t) == 44, "RTCI");
MLVFS_ABI_ASSERT(sizeof(mlv_idnt_hdr_t) == 84, "IDNT");
MLVFS_ABI_ASSERT(sizeof(mlv_wbal_hdr_t) == 44, "WBAL");
MLVFS_ABI_ASSERT(sizeof(struct frame_headers) == 592, "frame_headers");

#endif
