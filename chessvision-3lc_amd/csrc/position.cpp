// position.cpp -- (boards, 64, 13) class probabilities -> labels, FEN piece placement and the pawn rule, for a whole job.
//
// Host-side C++ counterpart of ChessVision.process_position_probabilities / validate_position / board_fen
// (reference chessvision/core.py:309-355, 441-469; python-chess' board_fen at core.py:330-349).  The batched pipeline
// decodes hundreds of boards per job; doing it per board in Python (argmax, list comprehensions, two FEN assemblies,
// an argsort) cost more host time than the GPU needed for the CNNs.  Same results as the Python restatement
// (chessvision/fen.py, core.py), which stays the readable version and the test checker.
#include <cstdint>
#include <cstring>

namespace cv {

static const char kLabels[14] = "BKNPQRbknpqrf";      // class index -> FEN symbol, "f" = empty (constants.LABEL_NAMES)
static const int kPawnW = 3, kPawnB = 9;

// labels[64] in classifier order (a8..h8, a7..h7, ..., a1..h1; reversed when flip) -> "rnbqkbnr/pppppppp/8/..." (<= 71 chars)
static void write_fen(const int8_t* labels, int flip, char* out) {
    char* o = out;
    for (int r = 0; r < 8; ++r) {                    // FEN rank 8 first
        int empty = 0;
        for (int f = 0; f < 8; ++f) {
            const int idx = flip ? 63 - (r * 8 + f) : r * 8 + f;
            const char c = kLabels[labels[idx]];
            if (c == 'f') { ++empty; continue; }
            if (empty) { *o++ = (char)('0' + empty); empty = 0; }
            *o++ = c;
        }
        if (empty) *o++ = (char)('0' + empty);
        if (r < 7) *o++ = '/';
    }
    *o = 0;
}

// fen / original_fen: n x 72 chars; labels: n x 64 validated class indices; fixes: up to 16 per board x
// {board, square index, original class, corrected class}; *n_fixes = number of fix records written.
void decode_positions(const float* probs, int n_boards, int flip, char* fen, char* original_fen, int8_t* labels,
                      int32_t* fixes, int32_t* n_fixes) {
    int nf = 0;
    for (int b = 0; b < n_boards; ++b) {
        const float* p = probs + (size_t)b * 64 * 13;
        int8_t* lab = labels + (size_t)b * 64;
        for (int i = 0; i < 64; ++i) {               // np.argmax: first maximum
            int best = 0;
            for (int k = 1; k < 13; ++k)
                if (p[i * 13 + k] > p[i * 13 + best]) best = k;
            lab[i] = (int8_t)best;
        }
        write_fen(lab, flip, original_fen + (size_t)b * 72);
        // rule "no_pawns_on_ends" (the only live rule, core.py:453-469): a pawn on rank 1 or 8 becomes the most probable
        // non-pawn class.  Ranks 8 and 1 are the first and last eight squares in either orientation.
        for (int i = 0; i < 64; ++i) {
            if (i >= 8 && i < 56) continue;
            if (lab[i] != kPawnW && lab[i] != kPawnB) continue;
            int alt = -1;                            // reversed stable ascending argsort: among equal values the higher index first
            for (int k = 0; k < 13; ++k) {
                if (k == kPawnW || k == kPawnB) continue;
                if (alt < 0 || p[i * 13 + k] >= p[i * 13 + alt]) alt = k;
            }
            fixes[nf * 4 + 0] = b; fixes[nf * 4 + 1] = i; fixes[nf * 4 + 2] = lab[i]; fixes[nf * 4 + 3] = alt;
            ++nf;
            lab[i] = (int8_t)alt;
        }
        write_fen(lab, flip, fen + (size_t)b * 72);
    }
    *n_fixes = nf;
}

}  // namespace cv
