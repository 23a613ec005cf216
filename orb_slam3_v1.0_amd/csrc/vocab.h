// vocab.h -- vocabulary-tree descent (DBoW2 TemplatedVocabulary::transform) on the device.
#pragma once
#include <string>
#include <vector>

#include "orbfe_internal.h"

namespace orbfe {

struct Vocab {
    int nNodes = 0, L = 0, maxChildren = 0;
    int* dChildOff = nullptr;   // [nNodes + 1]
    int* dChildIdx = nullptr;   // CSR: children of node i in DBoW2 order
    uint8_t* dDesc = nullptr;   // [nNodes][32]
    int* dWordId = nullptr;     // [nNodes]
    double* dWeight = nullptr;  // [nNodes]
    std::vector<double> hWeight; // host copy: the weight of the reached leaf is looked up on the host
    // per-call scratch (grown on demand)
    uint8_t* dIn = nullptr;
    int* dOut = nullptr;
    size_t cap = 0;
    void* hpin = nullptr;
};

int vocab_create(int nNodes, const int* childOff, const int* childIdx, const uint8_t* nodeDesc, const int* wordId,
                 const double* weight, int L, hipStream_t s, Vocab** out, std::string& err);
void vocab_destroy(Vocab* v);
int vocab_transform(Vocab* v, hipStream_t s, const uint8_t* desc, int n, int levelsup, int* wordOut, int* nodeOut,
                    double* weightOut, std::string& err);
// device descriptors, device count (<= cap); dOut [cap][2] = (word id, node id), dLeaf [cap] = the leaf reached
// dClear (optional, [cap]): set to -1 by the same launch
int vocab_transform_launch_dev(const Vocab* v, hipStream_t s, const uint8_t* dDesc, const int* dN, int cap, int levelsup,
                               int* dOut, int* dLeaf, int* dClear, std::string& err);

}  // namespace orbfe
