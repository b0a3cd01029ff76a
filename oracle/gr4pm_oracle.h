/*
 * gr4pm_oracle.h -- CPU oracle for the gr4-packet-modem RX hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and there only as the checker / reported baseline.
 *
 * This is a from-scratch restatement (plain C++17, no dependencies) of the
 * reference algorithms; every function cites the reference file:line it follows
 * (paths relative to /root/reference/blocks/include/gnuradio-4.0/packet-modem/).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *  - firdes / pfb_arb_taps / packet_transmitter_rrc_taps: pinned bit-exact against the
 *    reference's own standalone headers compiled into oracle/_ref/ (oracle/Makefile)
 *    and against the 65-tap literal of test/qa_firdes.cpp:11-34.
 *  - every block: pinned against the known-answer expectations of the reference's own
 *    tests (test/qa_*.cpp), restated in tests/test_oracle_*.py with the same stimulus
 *    and the same assertions.
 *  - the FFT inside SyncwordDetection is FFTW3f reached through gnuradio4's
 *    gr::algorithm::FFTw (git submodule `gnuradio4`, no pinned commit visible, absent
 *    from /root/reference).  Its published contract (forward sign, un-normalised,
 *    out-of-place) is restated here as a Stockham radix-4 float32 FFT; raw FFT bits are
 *    NOT pinned by any reference test, so FFT-derived float values are compared with a
 *    tolerance while indices / freq_bin / pass-through are compared exactly.
 *  - pinned by the restated tolerance bands of the reference's tests plus a reading of the reference code only
 *    (no reference test or reference-built binary pins them more tightly): the bits of the FFT and of the
 *    template spectra (the product computes its templates with a double-precision FFT rounded once: nothing
 *    pins template bits), SymbolFilter WITH tags (arm selection, clock-phase special cases, tag re-timing:
 *    symbol_filter.hpp:160-228), CoarseFrequencyCorrection with delay > 0, the float VALUES of the detector's
 *    tags beyond the bands of qa_syncword_detection.cpp:121-147, PfbArbResampler with a float rate.
 *  - CostasLoop's cosf / sinf are the host libm's: the product restates glibc's algorithm on the device and is
 *    pinned against this libm for every float of the loop's phase range (tests/sincosf_glibc_check.c).
 *  - the gr::Block runtime (gnuradio4) is absent, so the reference blocks themselves
 *    cannot be built here without writing stand-ins: no reference block executable exists.
 *  - cross-check that pins NOTHING (the API under the blocks is a stand-in) but catches restatement slips: in
 *    the build container tests/test_gr4_blocks.py compiles the reference's own rotator / coarse_frequency_
 *    correction / symbol_filter / costas_loop / interpolating_fir_filter / pfb_arb_resampler / syncword_wipeoff
 *    headers against the test stand-in tests/gr4_stub/ and compares their outputs with this oracle on exactly
 *    the paths listed above as pinned by code reading only (tags into SymbolFilter, CFC delay 26, float-rate
 *    resampler, phase tags into the Costas loop): bit-identical.  syncword_detection.hpp itself is driven the same
 *    way on orc_fft (below) as its gr::algorithm::FFTw: its items, tag positions and tag values equal orc_sd_process's
 *    bit for bit, i.e. the detector scan (:267-298) and output_tag (:56-115) are restated without a slip.
 */
#ifndef GR4PM_ORACLE_H
#define GR4PM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_c64;

/* syncword_detection.hpp:106-114 -- the seven tag keys, plus the output index */
typedef struct {
    uint64_t index;        /* absolute output item index the tag is attached to */
    float amplitude;       /* syncword_amplitude */
    float phase;           /* syncword_phase */
    double freq;           /* syncword_freq */
    int32_t freq_bin;      /* syncword_freq_bin */
    float noise_power;     /* syncword_noise_power */
    float esn0_db;         /* syncword_esn0_db */
    float time_est;        /* syncword_time_est */
    int32_t flags;         /* bit0: carries syncword_* keys; bit1: carries other keys */
    int32_t user;          /* the caller's cookie of gr4pm_tag (include/gr4pm_hip.h): never read, copied with the tag */
} orc_tag;

/* firdes.hpp:29-76; returns number of taps written (ntaps|1) */
size_t orc_rrc_taps(double gain, double fs, double symbol_rate, double alpha, size_t ntaps,
                    float* out);
/* packet_transmitter_rrc_taps.hpp:8-28 */
size_t orc_tx_rrc_taps(size_t sps, float* out);
/* forward, un-normalised complex FFT, n a power of two >= 2 (FFTW contract restated) */
void orc_fft(const orc_c64* in, orc_c64* out, size_t n);

/* ---- SyncwordDetection (syncword_detection.hpp:32-357) ---- */
typedef struct orc_sd orc_sd;
orc_sd* orc_sd_create(size_t fft_size, size_t sps, const float* rrc_taps, size_t n_taps,
                      const uint8_t* syncword, size_t n_syncword, const orc_c64* constellation,
                      size_t n_constellation, int min_freq_bin, int max_freq_bin,
                      uint64_t time_threshold, float power_threshold);
void orc_sd_destroy(orc_sd*);
size_t orc_sd_syncword_samples_size(const orc_sd*);
float orc_sd_self_corr(const orc_sd*);
/* copies template bin `b` (conj FFT of shifted syncword), fft_size values */
void orc_sd_template(const orc_sd*, size_t b, orc_c64* out);
/* processBulk :204-356.  Returns 0 OK, 1 INSUFFICIENT_INPUT_ITEMS.  *n_done = items
 * consumed == published.  zpow_dbg (optional, n_in floats) receives the per-sample
 * best-bin correlation power, bin_dbg (optional) the best bin index. */
int orc_sd_process(orc_sd*, const orc_c64* in, size_t n_in, orc_c64* out, size_t* n_done,
                   orc_tag* tags, size_t tags_cap, size_t* n_tags, float* zpow_dbg,
                   int32_t* bin_dbg);

/* ---- SyncwordDetectionFilter (syncword_detection_filter.hpp:54-210) ----
 * One call == one processBulk call.  tag_flags: 0 no tag at in[0], else orc_tag.flags.
 * n_headers / header_*: pending parsed_header messages (packet_length, or invalid);
 * n_ignored: pending ignored_syncword messages.  Outputs consumed counts. */
typedef struct orc_sdf orc_sdf;
orc_sdf* orc_sdf_create(size_t sps, size_t syncword_size, size_t header_size);
void orc_sdf_destroy(orc_sdf*);
int orc_sdf_process(orc_sdf*, const orc_c64* in, size_t n_in, orc_c64* out, size_t out_cap,
                    int tag_flags, size_t n_headers, const uint64_t* header_packet_length,
                    const uint8_t* header_invalid, size_t n_ignored, size_t* consumed,
                    size_t* header_consumed, size_t* ignored_consumed, int* tag_out_flags);

/* ---- CoarseFrequencyCorrection (coarse_frequency_correction.hpp:50-98) ----
 * Whole-stream driver: tags sorted by index; the stream is cut at tag indices exactly as
 * the runtime presents one tag at the head of a chunk. */
typedef struct orc_cfc orc_cfc;
orc_cfc* orc_cfc_create(size_t delay);
void orc_cfc_destroy(orc_cfc*);
void orc_cfc_process(orc_cfc*, const orc_c64* in, size_t n, orc_c64* out,
                     const uint64_t* tag_index, const double* tag_freq, size_t n_tags);

/* ---- Rotator (rotator.hpp:44-65) ---- */
typedef struct orc_rot orc_rot;
orc_rot* orc_rot_create(float phase_incr);
void orc_rot_destroy(orc_rot*);
void orc_rot_process(orc_rot*, const orc_c64* in, size_t n, orc_c64* out);

/* ---- CostasLoop (costas_loop.hpp:52-148); constellation 0 PILOT 1 BPSK 2 QPSK ---- */
typedef struct orc_costas orc_costas;
orc_costas* orc_costas_create(double loop_bandwidth, int constellation);
void orc_costas_destroy(orc_costas*);
void orc_sincosf(const float* x, size_t n, float* s, float* c);
void orc_costas_coeffs(const orc_costas*, float* k1, float* k2);
void orc_costas_process(orc_costas*, const orc_c64* in, size_t n, orc_c64* out,
                        const uint64_t* tag_index, const float* tag_phase, size_t n_tags);

/* ---- SyncwordWipeoff (syncword_wipeoff.hpp:38-90) ---- */
typedef struct orc_wipe orc_wipe;
orc_wipe* orc_wipe_create(const float* syncword, size_t n);
void orc_wipe_destroy(orc_wipe*);
void orc_wipe_process(orc_wipe*, const orc_c64* in, size_t n, orc_c64* out,
                      const uint64_t* tag_index, size_t n_tags);

/* ---- InterpolatingFirFilter (interpolating_fir_filter.hpp:42-102) ---- */
typedef struct orc_ifir orc_ifir;
orc_ifir* orc_ifir_create(size_t interpolation, const float* taps, size_t n_taps);
void orc_ifir_destroy(orc_ifir*);
void orc_ifir_process_c64(orc_ifir*, const orc_c64* in, size_t n, orc_c64* out);
void orc_ifir_process_f32(orc_ifir*, const float* in, size_t n, float* out);
/* stateless integer instantiation used by test/qa_interpolating_fir_filter.cpp */
void orc_ifir_int(size_t interpolation, const int* taps, size_t n_taps, const int* in, size_t n,
                  int* out);

/* ---- SymbolFilter (symbol_filter.hpp:64-252) ----
 * Whole-stream driver; tags (sorted) carry amplitude/time_est/phase/freq and flags.
 * Output tags are re-timed to symbol indices (:204-228); returns produced symbols. */
typedef struct orc_symf orc_symf;
orc_symf* orc_symf_create(size_t sps, const float* taps, size_t n_taps, size_t num_arms,
                          size_t delay);
void orc_symf_destroy(orc_symf*);
size_t orc_symf_process_c64(orc_symf*, const orc_c64* in, size_t n, orc_c64* out, size_t out_cap,
                            const orc_tag* tags_in, size_t n_tags_in, orc_tag* tags_out,
                            size_t tags_cap, size_t* n_tags_out, size_t* consumed);
size_t orc_symf_process_f32(orc_symf*, const float* in, size_t n, float* out, size_t out_cap,
                            size_t* consumed);

/* ---- PfbArbResampler (pfb_arb_resampler.hpp:67-182); rate_is_double selects TRate ---- */
typedef struct orc_arb orc_arb;
orc_arb* orc_arb_create(double rate, int rate_is_double, const float* taps, size_t n_taps,
                        size_t filter_size);
void orc_arb_destroy(orc_arb*);
size_t orc_arb_process(orc_arb*, const orc_c64* in, size_t n, orc_c64* out, size_t out_cap,
                       size_t* consumed);

/* ---- symbol-rate control tags behind SyncwordWipeoff (payload_metadata_insert.hpp:44-51) ----
 * kind 1: start of syncword ("constellation": PILOT, "loop_bandwidth", + the syncword_* keys)
 * kind 2: start of header   ("constellation": QPSK, "header_start", "loop_bandwidth")
 * kind 3: start of payload  (the parsed header + "payload_symbols", "payload_bits",
 *         "loop_bandwidth"; no "constellation" key)
 * constellation < 0 / loop_bandwidth < 0: key absent. */
typedef struct {
    uint64_t index;
    int32_t kind;
    int32_t constellation;
    double loop_bandwidth;
    uint64_t packet_length;
    uint64_t payload_symbols;
    uint64_t payload_bits;
    orc_tag syncword;
} orc_ptag;

/* ---- PayloadMetadataInsert (payload_metadata_insert.hpp:77-307) ----
 * Whole-stream driver: the stream is cut at the (sorted) syncword tags exactly as the runtime
 * presents one tag at the head of a chunk; headers = pending parsed_header messages in order.
 * Stops (consumed < n_in) where the block would wait for a header that is not there. */
typedef struct orc_pmi orc_pmi;
orc_pmi* orc_pmi_create(size_t syncword_size, size_t header_size, double syncword_bw,
                        double header_bw, double payload_bw);
void orc_pmi_destroy(orc_pmi*);
int orc_pmi_process(orc_pmi*, const orc_c64* in, size_t n_in, orc_c64* out, size_t out_cap,
                    const orc_tag* tags_in, size_t n_tags_in, const uint64_t* header_packet_length,
                    const uint8_t* header_invalid, size_t n_headers, orc_ptag* tags_out,
                    size_t tags_cap, size_t* n_tags_out, size_t* consumed, size_t* produced,
                    size_t* headers_used, size_t* ignored_syncwords);

/* ---- CostasLoop driven by the control tags: "constellation" / "loop_bandwidth" update the
 * settings (costas_loop.hpp:52-88), "syncword_phase" sets the phase (:101-106) ---- */
void orc_costas_process_packets(orc_costas*, const orc_c64* in, size_t n, orc_c64* out,
                                const orc_ptag* tags, size_t n_tags);

/* ---- SyncwordRemove (syncword_remove.hpp:39-105): whole-stream driver ---- */
typedef struct orc_sr orc_sr;
orc_sr* orc_sr_create(size_t syncword_size);
void orc_sr_destroy(orc_sr*);
size_t orc_sr_process(orc_sr*, const orc_c64* in, size_t n, orc_c64* out, const orc_ptag* tags_in,
                      size_t n_tags_in, orc_ptag* tags_out, size_t tags_cap, size_t* n_tags_out);
/* the int instantiation test/qa_syncword_remove.cpp uses (tags: kind 1 at tag_index[]) */
size_t orc_sr_process_int(orc_sr*, const int* in, size_t n, int* out, const uint64_t* tag_index,
                          size_t n_tags);

/* ---- ConstellationLLRDecoder (constellation_llr_decoder.hpp:55-134); returns LLRs written,
 * or (size_t)-1 for a constellation the block rejects (:72-74) ---- */
typedef struct orc_llr orc_llr;
orc_llr* orc_llr_create(float noise_sigma, int constellation);
void orc_llr_destroy(orc_llr*);
size_t orc_llr_process(orc_llr*, const orc_c64* in, size_t n, float* out, const orc_ptag* tags_in,
                       size_t n_tags_in, orc_ptag* tags_out, size_t tags_cap, size_t* n_tags_out);

/* ==== header decode loop (packet_receiver.hpp:131-139) ==== */

/* ---- AdditiveScrambler<float|uint8_t> (additive_scrambler.hpp:58-100): LFSR (mask, seed,
 * length) as in GNU Radio 3.10; soft symbols change sign where the LFSR bit is 1, hard symbols
 * are XORed.  Reset by `count` items or at the items listed in reset_index (reset_tag_key). ---- */
typedef struct orc_scr orc_scr;
orc_scr* orc_scr_create(uint64_t mask, uint64_t seed, uint64_t length, uint64_t count);
void orc_scr_destroy(orc_scr*);
void orc_scr_process_f32(orc_scr*, const float* in, size_t n, float* out, const uint64_t* reset_index,
                         size_t n_resets);
void orc_scr_process_u8(orc_scr*, const uint8_t* in, size_t n, uint8_t* out, const uint64_t* reset_index,
                        size_t n_resets);

/* ---- HeaderPayloadSplit<float> (header_payload_split.hpp:38-135): whole-stream driver.
 * tags: orc_ptag; kind 3 carries "payload_bits".  Returns 0, or -1 for the "received
 * unexpected payload_bits tag" exception (:75-78). ---- */
typedef struct orc_hps orc_hps;
orc_hps* orc_hps_create(size_t header_size);
void orc_hps_destroy(orc_hps*);
int orc_hps_process(orc_hps*, const float* in, size_t n, float* header, size_t* n_header, float* payload,
                    size_t* n_payload, const orc_ptag* tags, size_t n_tags, orc_ptag* header_tags,
                    size_t* n_header_tags, orc_ptag* payload_tags, size_t* n_payload_tags, size_t tags_cap);

/* ---- HeaderFecEncoder (header_fec_encoder.hpp:60-107): 4 bytes -> 32 bytes; `generator` = the
 * 96 rows of the dense generator (tests/golden/header_ldpc_generator.npy) ---- */
void orc_header_fec_encode(const uint32_t* generator, const uint8_t* in, size_t n_codewords, uint8_t* out);

/* ---- HeaderFecDecoder (header_fec_decoder.hpp:290-347): 256 LLRs -> 4 bytes + invalid flag.
 * The LDPC decoder itself lives in the reference's Rust dependency ldpc-toolbox (absent from
 * /root/reference; decoder implementation "HLAminstari8", :276): restated here as the
 * horizontal-layered A-Min*-BP schedule in float32 with the correction term ln(1+e^-x) tabulated
 * in steps of 1/8 -- NOT its 8-bit arithmetic: parity with the reference decoder is pinned only
 * on the reference's own QA vectors (valid codewords decode, random words are rejected). ---- */
typedef struct orc_ldpc orc_ldpc;
orc_ldpc* orc_ldpc_create(const char* alist);
void orc_ldpc_destroy(orc_ldpc*);
/* returns iterations used (0 = the input already was a codeword), -1 = no codeword found */
int orc_ldpc_decode(orc_ldpc*, const float* llrs, uint8_t* bits_k, unsigned max_iterations);
/* the product's second message arithmetic (8-bit messages; include/gr4pm_hip.h), same return values */
int orc_ldpc_decode_q8(orc_ldpc*, const float* llrs, uint8_t* bits_k, unsigned max_iterations);
void orc_header_fec_decode(orc_ldpc*, const float* llrs, size_t n_codewords, uint8_t* bytes,
                           uint8_t* invalid);
void orc_header_fec_decode_q8(orc_ldpc*, const float* llrs, size_t n_codewords, uint8_t* bytes, uint8_t* invalid);

/* ---- Crc<uint64_t> (crc.hpp:31-156) and CrcCheck (crc_check.hpp:75-216) ---- */
typedef struct {
    unsigned num_bits;
    uint64_t poly, initial_value, final_xor;
    int input_reflected, result_reflected;
} orc_crc_params;
uint64_t orc_crc_compute(const orc_crc_params*, const uint8_t* data, size_t n);
/* packets back to back in `in`, lengths in packet_len[]; passing packets are copied to `out`
 * (without the CRC when discard_crc); out_len[i] = bytes written for packet i (0: dropped).
 * Returns the number of output bytes. */
size_t orc_crc_check(const orc_crc_params*, int swap_endianness, int discard_crc, uint64_t skip_header_bytes,
                     const uint8_t* in, const uint64_t* packet_len, size_t n_packets, uint8_t* out,
                     uint64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
