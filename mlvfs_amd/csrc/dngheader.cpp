// dngheader.cpp -- the CinemaDNG header of a virtual .dng file.
//
// Replaces  mlvfs/dng.c:477-789  size_t dng_get_header_data(frame_headers*, uint8_t*, off_t, size_t, double, char*)
// with       mlvfs/dng.c:59-261   (per-camera colour calibration and focal-plane resolution)
//            mlvfs/dng.c:263-470  (Kelvin -> AsShotNeutral for the non-custom white-balance modes)
//
// Pure host code: 64 KiB of TIFF structure per frame, built from the MLV block headers; there is nothing here for
// a GPU to do.  It lives in this library because a replacement for dng.o has to export it and because
// header + pixels together make the byte-exact virtual file.  Checked byte for byte against the reference build
// (tests/test_header.py, tests/golden/header_cases.npz).
//
// Layout produced (all little endian, everything after the last value is zero up to 65536):
//     0  TIFF header  "II" 42, IFD0 at 8
//     8  IFD0: 41 entries, next = 0
//   506  EXIF IFD: 11 entries, next = 0
//   644  out-of-line values in the order the entries list them (strings NUL-terminated and padded to even length;
//        strings of <= 4 bytes including the NUL live in the entry itself)
// Reference behaviours kept on purpose:
//   * the active area of `frame_headers` is rewritten in the caller's struct when the recorded frame does not
//     contain the optical-black borders (dng.c:665-672)
//   * the float (not double) reciprocal of the pre-multipliers in the Kelvin path (dng.c:383), the unwrapped day of
//     format_datetime, ISO stored as a full 32-bit value in a SHORT entry, `(int32_t)fps_override * 1000`
//   * the returned size is min(max_size, 65536) whatever `offset` is (dng.c:779).  The reference then copies that many
//     bytes starting at header + offset, i.e. past the end of its buffer for offset > 0; here bytes past the
//     header read as zero.  MLVFS itself only ever asks for offset 0 (main.c:944,964).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mlvfs_amd.h"

namespace {

constexpr size_t kHeaderBytes = 65536;

// ---- calibration data (Adobe DNG Converter values for the Magic Lantern cameras; all denominators 10000) ----------
struct Calibration {
    const char *model;
    short m[4][9];            // ColorMatrix1 (StdA), ColorMatrix2 (D65), ForwardMatrix1, ForwardMatrix2
};
const Calibration kCalibration[] = {
    { "Canon EOS 5D Mark III",
      { {7234,-1413,-600,-3631,11150,2850,-382,1335,6437},
        {6722,-635,-963,-4287,12460,2028,-908,2162,5668},
        {7868,92,1683,2291,8615,-906,27,-4752,12976},
        {7637,805,1201,2649,9179,-1828,137,-2456,10570} } },
    { "Canon EOS 5D Mark II",
      { {5309,-229,-336,-6241,13265,3337,-817,1215,6664},
        {4716,603,-830,-7798,15474,2480,-1496,1937,6651},
        {8924,-1041,1760,4351,6621,-972,505,-1562,9308},
        {8924,-1041,1760,4351,6621,-972,505,-1562,9308} } },
    { "Canon EOS 7D",
      { {11620,-6350,5,-2558,10146,2813,24,858,6926},
        {6844,-996,-856,-3876,11761,2396,-593,1772,6198},
        {5445,3536,662,1106,10136,-1242,-374,-3559,12184},
        {7415,1533,695,2499,9997,-2497,-22,-1933,10207} } },
    { "Canon EOS 6D",
      { {7546,-1435,-929,-3846,11488,2692,-332,1209,6370},
        {7034,-804,-1014,-4420,12564,2058,-851,1994,5758},
        {7763,65,1815,2364,8351,-715,-59,-4228,12538},
        {7464,1044,1135,2648,9173,-1820,113,-2154,10292} } },
    { "Canon EOS 70D",
      { {7546,-1435,-929,-3846,11488,2692,-332,1209,6370},
        {7034,-804,-1014,-4420,12564,2058,-851,1994,5758},
        {7763,65,1815,2364,8351,-715,-59,-4228,12538},
        {7464,1044,1135,2648,9173,-1820,113,-2154,10292} } },
    { "Canon EOS 60D",
      { {7428,-1897,-491,-3505,10963,2929,-337,1242,6413},
        {6719,-994,-925,-4408,12426,2211,-887,2129,6051},
        {7550,645,1448,2138,8936,-1075,-5,-4306,12562},
        {7286,1385,972,2600,9468,-2068,93,-2268,10426} } },
    { "Canon EOS 50D",
      { {5852,-578,-41,-4691,11696,3427,-886,2323,6879},
        {4920,616,-593,-6493,13964,2784,-1774,3178,7005},
        {8716,-692,1618,3408,8077,-1486,-13,-6583,14847},
        {9485,-1150,1308,4313,7807,-2120,293,-2826,10785} } },
    { "Canon EOS 550D",
      { {7755,-2449,-349,-3106,10222,3362,-156,986,6409},
        {6941,-1164,-857,-3825,11597,2534,-416,1540,6039},
        {7163,1301,1179,1926,9543,-1469,-278,-3830,12359},
        {7239,1838,566,2467,10246,-2713,-112,-1754,10117} } },
    { "Canon EOS 600D",
      { {7164,-1916,-431,-3361,10600,3200,-272,1058,6442},
        {6461,-907,-882,-4300,12184,2378,-819,1944,5931},
        {7486,835,1322,2099,9147,-1245,-12,-3822,12085},
        {7359,1365,918,2610,9687,-2297,98,-2155,10309} } },
    { "Canon EOS 650D",
      { {6985,-1611,-397,-3596,10749,3295,-349,1136,6512},
        {6602,-841,-939,-4472,12458,2247,-975,2039,6148},
        {7747,485,1411,2340,8840,-1180,105,-4147,12293},
        {7397,1199,1047,2650,9355,-2005,193,-2113,10171} } },
    { "Canon EOS 700D",
      { {6985,-1611,-397,-3596,10749,3295,-349,1136,6512},
        {6602,-841,-939,-4472,12458,2247,-975,2039,6148},
        {7747,485,1411,2340,8840,-1180,105,-4147,12293},
        {7397,1199,1047,2650,9355,-2005,193,-2113,10171} } },
    { "Canon EOS 1100D",
      { {6873,-1696,-529,-3659,10795,3313,-362,1165,7234},
        {6444,-904,-893,-4563,12308,2535,-903,2016,6728},
        {7607,647,1389,2337,8876,-1213,93,-3625,11783},
        {7357,1377,909,2729,9630,-2359,104,-1940,10087} } },
    { "Canon EOS M",
      { {7357,1377,909,2729,9630,-2359,104,-1940,10087},
        {6602,-841,-939,-4472,12458,2247,-975,2039,6148},
        {7747,485,1411,2340,8840,-1180,105,-4147,12293},
        {7397,1199,1047,2650,9355,-2005,193,-2113,10171} } },
};

// sensor pixels per inch as a rational: {x numerator, x denominator, y numerator, y denominator}; unit is always inches (2)
struct FocalPlane { const char *model; int xn, xd, yn, yd; };
const FocalPlane kFocalPlane[] = {
    { "Canon EOS 5D Mark III", 5760000, 1461, 3840000, 972 }, { "Canon EOS 5D Mark II", 5616000, 1459, 3744000, 958 },
    { "Canon EOS 7D", 5184000, 907, 3456000, 595 },           { "Canon EOS 6D", 5472000, 1436, 3648000, 956 },
    { "Canon EOS 60D", 5184000, 905, 3456000, 595 },          { "Canon EOS 70D", 5472000, 899, 3648000, 599 },
    { "Canon EOS 50D", 4752000, 894, 3168000, 597 },          { "Canon EOS 500D", 4752000, 894, 3168000, 593 },
    { "Canon EOS 550D", 5184000, 905, 3456000, 595 },         { "Canon EOS 600D", 5184000, 905, 3456000, 595 },
    { "Canon EOS 650D", 5184000, 894, 3456000, 597 },         { "Canon EOS 700D", 5184000, 894, 3456000, 597 },
    { "Canon EOS 1100D", 4272000, 905, 2848000, 595 },        { "Canon EOS M", 5184000, 894, 3456000, 597 },
};

template <class T, size_t N>
const T &by_model(const T (&table)[N], const char *model)        // unknown cameras use the first row (dng.c:622-630,690-698)
{
    for (const T &e : table)
        if (!strcmp(e.model, model)) return e;
    return table[0];
}

// ---- Kelvin -> channel multipliers (dng.c:263-420; the fits are the CIE daylight locus as used by UFRaw) -------------
// Arithmetic is written operation for operation like the reference: doubles, except where the reference rounds
// through float (pre-multipliers and the inverted camera matrix).
void daylight_rgb(double T, double (&rgb)[3])
{
    static const double xyz2rgb[3][3] = { { 3.24071, -0.969258, 0.0556352 }, { -1.53726, 1.87599, -0.203996 }, { -0.498571, 0.0415557, 1.05707 } };
    double xD;
    if (T <= 4000) xD = 0.27475e9 / (T * T * T) - 0.98598e6 / (T * T) + 1.17444e3 / T + 0.145986;
    else if (T <= 7000) xD = -4.6070e9 / (T * T * T) + 2.9678e6 / (T * T) + 0.09911e3 / T + 0.244063;
    else xD = -2.0064e9 / (T * T * T) + 1.9018e6 / (T * T) + 0.24748e3 / T + 0.237040;
    const double yD = -3 * xD * xD + 2.87 * xD - 0.275;
    const double X = xD / yD, Y = 1, Z = (1 - xD - yD) / yD;
    double top = 0;
    for (int c = 0; c < 3; c++) {
        rgb[c] = X * xyz2rgb[0][c] + Y * xyz2rgb[1][c] + Z * xyz2rgb[2][c];
        if (rgb[c] > top) top = rgb[c];
    }
    for (int c = 0; c < 3; c++) rgb[c] = rgb[c] / top;
}

// out = in * (in^T in)^-1 for a rows x 3 matrix (Gauss-Jordan on the normal equations, no pivoting)
void pinv3(const double (*in)[3], double (*out)[3], int rows)
{
    double w[3][6];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 6; j++) w[i][j] = j == i + 3;
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < rows; k++) w[i][j] += in[k][i] * in[k][j];
    }
    for (int i = 0; i < 3; i++) {
        double d = w[i][i];
        for (int j = 0; j < 6; j++) w[i][j] /= d;
        for (int k = 0; k < 3; k++) {
            if (k == i) continue;
            d = w[k][i];
            for (int j = 0; j < 6; j++) w[k][j] -= w[i][j] * d;
        }
    }
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < 3; j++) {
            out[i][j] = 0;
            for (int k = 0; k < 3; k++) out[i][j] += w[j][k + 3] * in[i][k];
        }
}

void kelvin_to_neutral(double kelvin, const Calibration &cal, int32_t (&neutral)[6])
{
    static const double xyz_rgb[3][3] = { { 0.412453, 0.357580, 0.180423 }, { 0.212671, 0.715160, 0.072169 }, { 0.019334, 0.119193, 0.950227 } };
    double cam_xyz[3][3];
    for (int i = 0; i < 9; i++) cam_xyz[i / 3][i % 3] = (double)cal.m[1][i] / (double)10000;      // ColorMatrix2

    // camera <- sRGB, rows normalised to sum 1; the row sums' reciprocals are the pre-multipliers (kept as float)
    double cam_rgb[3][3], inv[3][3];
    float pre_mul[3], rgb_cam[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            cam_rgb[i][j] = 0;
            for (int k = 0; k < 3; k++) cam_rgb[i][j] += cam_xyz[i][k] * xyz_rgb[k][j];
        }
    for (int i = 0; i < 3; i++) {
        double sum = 0;
        for (int j = 0; j < 3; j++) sum += cam_rgb[i][j];
        for (int j = 0; j < 3; j++) cam_rgb[i][j] /= sum;
        pre_mul[i] = 1 / sum;
    }
    pinv3(cam_rgb, inv, 3);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) rgb_cam[i][j] = inv[j][i];

    double rgb_cam_t[3][3], back[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) rgb_cam_t[i][j] = rgb_cam[j][i];
    pinv3(rgb_cam_t, back, 3);

    double wb[3];
    daylight_rgb(kelvin, wb);
    wb[1] = wb[1] / 1.0;                                                                         // "green" is always 1
    double mul[3];
    for (int c = 0; c < 3; c++) {
        double acc = 0;
        for (int cc = 0; cc < 3; cc++) acc += 1 / pre_mul[c] * back[c][cc] * wb[cc];            // 1 / float: a float division
        mul[c] = 1 / acc;
    }
    mul[0] /= mul[1];
    mul[2] /= mul[1];
    mul[1] = 1;
    for (int c = 0; c < 3; c++) {
        neutral[2 * c] = 1000000;
        neutral[2 * c + 1] = (int32_t)(mul[c] * 1000000);
    }
}

void as_shot_neutral(const mlv_wbal_hdr_t &wb, const Calibration &cal, int32_t (&neutral)[6])      // dng.c:422-470
{
    enum { AUTO = 0, SUNNY = 1, CLOUDY = 2, TUNGSTEN = 3, FLUORESCENT = 4, FLASH = 5, CUSTOM = 6, SHADE = 8, KELVIN = 9 };
    if (wb.wb_mode == CUSTOM) {
        const int32_t v[6] = { (int32_t)wb.wbgain_r, (int32_t)wb.wbgain_g, (int32_t)wb.wbgain_g, (int32_t)wb.wbgain_g,
                               (int32_t)wb.wbgain_b, (int32_t)wb.wbgain_g };
        memcpy(neutral, v, sizeof v);
        return;
    }
    double kelvin = 5500;
    switch (wb.wb_mode) {
        case AUTO: case KELVIN: kelvin = wb.kelvin; break;
        case SHADE: kelvin = 7000; break;
        case CLOUDY: kelvin = 6000; break;
        case TUNGSTEN: kelvin = 3200; break;
        case FLUORESCENT: kelvin = 4000; break;
        default: break;                                       // sunny, flash and unknown modes: 5500 K
    }
    kelvin_to_neutral(kelvin, cal, neutral);
}

// ---- TIFF writer -------------------------------------------------------------------------------------------------
enum : uint16_t { BYTE = 1, ASCII = 2, SHORT = 3, LONG = 4, RATIONAL = 5, UNDEFINED = 7, SRATIONAL = 10 };

class Tiff {
public:
    Tiff(uint8_t *buf, size_t ifd0_entries, size_t exif_entries) : b_(buf)
    {
        const uint16_t head[4] = { 0x4949, 42, 8, 0 };
        memcpy(b_, head, 8);
        ifd_ = 8;
        exif_ifd = (uint32_t)(8 + 2 + 12 * ifd0_entries + 4);
        data_ = (uint32_t)(exif_ifd + 2 + 12 * exif_entries + 4);
        begin_ifd(ifd0_entries);
    }
    uint32_t exif_ifd;

    void begin_ifd(size_t entries)
    {
        put16(ifd_, (uint16_t)entries);
        ifd_ += 2;
        left_ = entries;
    }
    void end_ifd()                      // next-IFD offset 0 (already zero)
    {
        ifd_ += 4;
    }
    void inl(uint16_t tag, uint16_t type, uint32_t count, uint32_t value) { entry(tag, type, count, value); }
    void ascii(uint16_t tag, const char *s)
    {
        const size_t n = strlen(s) + 1;
        uint32_t v = 0;
        if (n <= 4) memcpy(&v, s, n);
        else {
            v = data_;
            if (data_ + n <= kHeaderBytes) memcpy(b_ + data_, s, n);
            data_ += (uint32_t)n;
            data_ += data_ & 1;
        }
        entry(tag, ASCII, (uint32_t)n, v);
    }
    void words(uint16_t tag, uint16_t type, uint32_t count, const int32_t *v, size_t nwords)
    {
        const uint32_t at = data_;
        if (data_ + 4 * nwords <= kHeaderBytes) memcpy(b_ + data_, v, 4 * nwords);
        data_ += (uint32_t)(4 * nwords);
        entry(tag, type, count, at);
    }
    void rational(uint16_t tag, uint16_t type, int32_t num, int32_t den)
    {
        const int32_t v[2] = { num, den };
        words(tag, type, 1, v, 2);
    }
    void bytes8(uint16_t tag, const uint8_t (&v)[8])
    {
        const uint32_t at = data_;
        memcpy(b_ + data_, v, 8);
        data_ += 8;
        entry(tag, BYTE, 8, at);
    }

private:
    void put16(size_t at, uint16_t v) { memcpy(b_ + at, &v, 2); }
    void entry(uint16_t tag, uint16_t type, uint32_t count, uint32_t value)
    {
        if (!left_) return;
        left_--;
        put16(ifd_, tag);
        put16(ifd_ + 2, type);
        memcpy(b_ + ifd_ + 4, &count, 4);
        memcpy(b_ + ifd_ + 8, &value, 4);
        ifd_ += 12;
    }
    uint8_t *b_;
    size_t ifd_ = 0, left_ = 0;
    uint32_t data_ = 0;
};

uint8_t bcd(int v) { return (uint8_t)(((v / 10) << 4) | (v % 10)); }

}  // namespace

extern "C" size_t dng_get_header_data(struct frame_headers *fh, uint8_t *output_buffer, off_t offset, size_t max_size,
                                      double fps_override, char *mlv_basename)
{
    std::vector<uint8_t> header(kHeaderBytes, 0);
    struct raw_info &ri = fh->rawi_hdr.raw_info;

    // camera identity: the model is the NUL-terminated cameraName as it sits in the struct, the make its first word
    const char *model = (const char *)fh->idnt_hdr.cameraName;
    char make[33];
    strncpy(make, model, 32);
    make[32] = 0;
    if (char *sp = strchr(make, ' ')) *sp = 0;
    char serial[33];
    memcpy(serial, fh->idnt_hdr.cameraSerial, 32);
    serial[32] = 0;

    // pixel aspect and focal-plane resolution: 5x3 line skipping when the sensor area is wider than 2:1 and at most
    // 720 rows, 3x3 binning when it is narrower than 2000 columns (dng.c:632-661)
    const FocalPlane &fp = by_model(kFocalPlane, model);
    int32_t fpx[2] = { fp.xn, fp.xd }, fpy[2] = { fp.yn, fp.yd };
    int32_t scale[4] = { 1, 1, 1, 1 };
    const double raw_w = ri.active_area.x2 - ri.active_area.x1, raw_h = ri.active_area.y2 - ri.active_area.y1;
    if (raw_w / raw_h > 2.0 && raw_h <= 720) {
        scale[2] = 5; scale[3] = 3;
        fpx[1] *= 3; fpy[1] *= 5;
    } else if (raw_w < 2000) {
        fpx[1] *= 3; fpy[1] *= 3;
    }
    // the recorded frame may not contain the optical-black borders the active area was measured against
    if (fh->rawi_hdr.xRes < ri.active_area.x2 || fh->rawi_hdr.yRes < ri.active_area.y2) {
        ri.active_area.x1 = 0; ri.active_area.y1 = 0;
        ri.active_area.x2 = fh->rawi_hdr.xRes; ri.active_area.y2 = fh->rawi_hdr.yRes;
    }

    int32_t rate[2] = { (int32_t)fh->file_hdr.sourceFpsNom, (int32_t)fh->file_hdr.sourceFpsDenom };
    if (fps_override > 0) { rate[0] = (int32_t)fps_override * 1000; rate[1] = 1000; }
    const double fps = rate[1] == 0 ? 0 : (double)rate[0] / (double)rate[1];
    int32_t exposure_bias[2] = { ri.exposure_bias[0], ri.exposure_bias[1] };
    if (exposure_bias[1] == 0) { exposure_bias[0] = 0; exposure_bias[1] = 1; }

    // wall-clock time of the frame: RTCI plus the microseconds between the RTCI and VIDF blocks (days do not wrap)
    char datetime[64];
    {
        const mlv_rtci_hdr_t &t = fh->rtci_hdr;
        const uint32_t sec = t.tm_sec + (uint32_t)((fh->vidf_hdr.timestamp - t.timestamp) / 1000000);
        const uint32_t min = t.tm_min + sec / 60, hour = t.tm_hour + min / 60, day = t.tm_mday + hour / 24;
        snprintf(datetime, sizeof datetime, "%04d:%02d:%02d %02d:%02d:%02d", 1900 + t.tm_year, t.tm_mon + 1, day, hour % 24,
                 min % 60, sec % 60);
    }
    // SMPTE time code from the frame number, at the frame rate rounded to an integer (dng.c:543-581)
    uint8_t timecode[8] = { 0 };
    {
        const int frame = (int)fh->vidf_hdr.frameNumber;
        const double t = fps == 0 ? 0 : frame / (fps > 1 ? round(fps) : fps);
        const int hh = (int)floor(t / 3600), mm = ((int)floor(t / 60)) % 60, ss = ((int)floor(t)) % 60;
        const int ff = fps > 1 ? (frame % ((int)round(fps))) : 0;
        timecode[0] = bcd(ff) & 0x3F; timecode[1] = bcd(ss) & 0x7F; timecode[2] = bcd(mm) & 0x7F; timecode[3] = bcd(hh) & 0x3F;
    }

    const Calibration &cal = by_model(kCalibration, model);
    int32_t matrix[4][18];
    for (int m = 0; m < 4; m++)
        for (int i = 0; i < 9; i++) { matrix[m][2 * i] = cal.m[m][i]; matrix[m][2 * i + 1] = 10000; }
    int32_t neutral[6];
    as_shot_neutral(fh->wbal_hdr, cal, neutral);

    Tiff t(header.data(), 41, 11);
    t.inl(254, LONG, 1, 0);                                             // NewSubFileType: main image
    t.inl(256, LONG, 1, fh->rawi_hdr.xRes);                             // ImageWidth
    t.inl(257, LONG, 1, fh->rawi_hdr.yRes);                             // ImageLength
    t.inl(258, SHORT, 1, 16);                                           // BitsPerSample
    t.inl(259, SHORT, 1, 1);                                            // Compression: none
    t.inl(262, SHORT, 1, 32803);                                        // PhotometricInterpretation: CFA
    t.inl(266, SHORT, 1, 1);                                            // FillOrder
    t.ascii(271, make);                                                 // Make
    t.ascii(272, model);                                                // Model
    t.inl(273, LONG, 1, (uint32_t)kHeaderBytes);                        // StripOffsets
    t.inl(274, SHORT, 1, 1);                                            // Orientation
    t.inl(277, SHORT, 1, 1);                                            // SamplesPerPixel
    t.inl(278, SHORT, 1, fh->rawi_hdr.yRes);                            // RowsPerStrip
    t.inl(279, LONG, 1, (uint32_t)dng_get_image_size(fh));              // StripByteCounts
    t.inl(284, SHORT, 1, 1);                                            // PlanarConfiguration
    t.ascii(305, "MLVFS");                                              // Software (mlvfs.h:65)
    t.ascii(306, datetime);                                             // DateTime
    t.inl(33421, SHORT, 2, 0x00020002);                                 // CFARepeatPatternDim 2x2
    t.inl(33422, BYTE, 4, 0x02010100);                                  // CFAPattern RGGB
    t.inl(34665, LONG, 1, t.exif_ifd);                                  // ExifIFD
    t.inl(50706, BYTE, 4, 0x00000401);                                  // DNGVersion 1.4.0.0
    t.ascii(50708, model);                                              // UniqueCameraModel
    t.inl(50714, LONG, 1, (uint32_t)ri.black_level);                    // BlackLevel
    t.inl(50717, LONG, 1, (uint32_t)ri.white_level);                    // WhiteLevel
    t.words(50718, RATIONAL, 2, scale, 4);                              // DefaultScale
    t.inl(50719, SHORT, 2, ((uint32_t)(uint16_t)ri.crop.origin[1] << 16) | (uint16_t)ri.crop.origin[0]);       // DefaultCropOrigin
    t.inl(50720, SHORT, 2, ((uint32_t)(uint16_t)(ri.active_area.y2 - ri.active_area.y1) << 16) |
                               (uint16_t)(ri.active_area.x2 - ri.active_area.x1));                                  // DefaultCropSize
    t.words(50721, SRATIONAL, 9, matrix[0], 18);                        // ColorMatrix1
    t.words(50722, SRATIONAL, 9, matrix[1], 18);                        // ColorMatrix2
    t.words(50728, RATIONAL, 3, neutral, 6);                            // AsShotNeutral
    t.words(50730, SRATIONAL, 1, exposure_bias, 2);                     // BaselineExposure
    t.ascii(50735, serial);                                             // CameraSerialNumber
    t.inl(50778, SHORT, 1, 17);                                         // CalibrationIlluminant1: standard light A
    t.inl(50779, SHORT, 1, 21);                                         // CalibrationIlluminant2: D65
    t.words(50829, LONG, 4, ri.dng_active_area, 4);                     // ActiveArea
    t.words(50964, SRATIONAL, 9, matrix[2], 18);                        // ForwardMatrix1
    t.words(50965, SRATIONAL, 9, matrix[3], 18);                        // ForwardMatrix2
    t.bytes8(51043, timecode);                                          // TimeCodes (CinemaDNG)
    t.words(51044, SRATIONAL, 1, rate, 2);                              // FrameRate (CinemaDNG)
    t.ascii(51081, mlv_basename ? mlv_basename : "");                   // ReelName (CinemaDNG)
    t.rational(51109, SRATIONAL, 0, 1);                                 // BaselineExposureOffset
    t.end_ifd();

    t.begin_ifd(11);
    t.rational(33434, RATIONAL, (int32_t)fh->expo_hdr.shutterValue / 1000, 1000);     // ExposureTime
    t.rational(33437, RATIONAL, fh->lens_hdr.aperture, 100);            // FNumber
    t.inl(34855, SHORT, 1, fh->expo_hdr.isoValue);                      // ISOSpeedRatings
    t.inl(34864, SHORT, 1, 3);                                          // SensitivityType: ISO speed
    t.inl(36864, UNDEFINED, 4, 0x30333230);                             // ExifVersion "0230"
    t.rational(37382, RATIONAL, fh->lens_hdr.focalDist, 1);             // SubjectDistance
    t.rational(37386, RATIONAL, fh->lens_hdr.focalLength, 1);           // FocalLength
    t.words(41486, RATIONAL, 1, fpx, 2);                                // FocalPlaneXResolution
    t.words(41487, RATIONAL, 1, fpy, 2);                                // FocalPlaneYResolution
    t.inl(41488, SHORT, 1, 2);                                          // FocalPlaneResolutionUnit: inches
    t.ascii(42036, (const char *)fh->lens_hdr.lensName);                // LensModel
    t.end_ifd();

    const size_t room = kHeaderBytes - (size_t)(offset < 0 ? offset : 0);             // dng.c:779: MIN(0, offset), not MAX
    const size_t n = max_size < room ? max_size : room;
    if (n) {
        size_t have = 0;
        if (offset >= 0 && (size_t)offset < kHeaderBytes) have = kHeaderBytes - (size_t)offset < n ? kHeaderBytes - (size_t)offset : n;
        if (have) memcpy(output_buffer, header.data() + offset, have);
        if (have < n) memset(output_buffer + have, 0, n - have);
    }
    return n;
}
