/*
 * orb_oracle_bow.c -- CPU ORACLE (test infrastructure, NOT product code; see orb_oracle.h).
 *
 * Restates the vocabulary-tree path ORB_SLAM2 takes right after extraction:
 *   Frame::ComputeBoW                      src/Frame.cc:395-402  (levelsup = 4)
 *   TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 *                                          Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1140-1207
 *   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)   :1231-1274
 *   FORB::distance                         Thirdparty/DBoW2/DBoW2/FORB.cpp:81-102
 *   BowVector::addWeight / addIfNotExist / normalize      BowVector.cpp:34-90
 *   FeatureVector::addFeature              FeatureVector.cpp:32-46
 *   loadFromTextFile's node / children / word numbering   TemplatedVocabulary.h:1348-1437
 * and the node-wise matcher ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) src/ORBmatcher.cc:159-288.
 *
 * DBoW2 is vendored in the reference tree, so this part is read, not recalled; what is missing is the vocabulary
 * FILE (Vocabulary/ORBvoc.bin is a missing blob): tests drive it with seeded synthetic trees.  No reference-held
 * fixture pins it either -> parity unpinned like the rest of the oracle.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct ora_vocabulary {
    int k, L, n_nodes, n_words;
    int weighting, scoring;
    int *parent;
    int *child_start; /* CSR over nodes: children in ascending node id = push_back order of the loader */
    int *child_items;
    uint8_t *desc;
    double *weight;
    int *word_id; /* -1 for inner nodes; leaves numbered in node order (:1421-1426) */
};

ora_vocabulary *ora_vocabulary_create(int k, int L, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                                      const uint8_t *desc, const double *weight, int weighting, int scoring)
{
    if (n_nodes < 1 || k < 1 || L < 1)
        return NULL;
    ora_vocabulary *v = (ora_vocabulary *)calloc(1, sizeof(*v));
    v->k = k;
    v->L = L;
    v->n_nodes = n_nodes;
    v->weighting = weighting;
    v->scoring = scoring;
    v->parent = (int *)malloc(sizeof(int) * (size_t)n_nodes);
    v->child_start = (int *)calloc((size_t)n_nodes + 1, sizeof(int));
    v->child_items = (int *)malloc(sizeof(int) * (size_t)n_nodes);
    v->desc = (uint8_t *)malloc((size_t)n_nodes * 32);
    v->weight = (double *)malloc(sizeof(double) * (size_t)n_nodes);
    v->word_id = (int *)malloc(sizeof(int) * (size_t)n_nodes);
    memcpy(v->desc, desc, (size_t)n_nodes * 32);
    memcpy(v->weight, weight, sizeof(double) * (size_t)n_nodes);
    v->parent[0] = -1;
    v->word_id[0] = -1;
    for (int i = 1; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] >= i) { /* the loader appends a node after its parent */
            ora_vocabulary_destroy(v);
            return NULL;
        }
        v->parent[i] = parent[i];
        v->child_start[parent[i] + 1]++;
    }
    for (int i = 0; i < n_nodes; i++)
        v->child_start[i + 1] += v->child_start[i];
    int *pos = (int *)malloc(sizeof(int) * (size_t)n_nodes);
    memcpy(pos, v->child_start, sizeof(int) * (size_t)n_nodes);
    for (int i = 1; i < n_nodes; i++)
        v->child_items[pos[parent[i]]++] = i;
    free(pos);
    int nw = 0;
    for (int i = 1; i < n_nodes; i++)
        v->word_id[i] = is_leaf[i] ? nw++ : -1;
    v->n_words = nw;
    return v;
}

void ora_vocabulary_destroy(ora_vocabulary *v)
{
    if (!v)
        return;
    free(v->parent);
    free(v->child_start);
    free(v->child_items);
    free(v->desc);
    free(v->weight);
    free(v->word_id);
    free(v);
}

int ora_vocabulary_words(const ora_vocabulary *v) { return v->n_words; }

/* TemplatedVocabulary.h:1231-1274.  Node::isLeaf() is children.empty().  If a leaf is met above the level the
 * node id is asked for, the reference leaves *nid untouched (an uninitialised local of the caller, :1163-1168);
 * the oracle reports the leaf itself there (documented convention). */
void ora_vocabulary_transform_feature(const ora_vocabulary *v, const uint8_t *feature, int levelsup, int32_t *word_id,
                                      double *weight, int32_t *nid)
{
    const int nid_level = v->L - levelsup;
    int got_nid = 0;
    if (nid_level <= 0) {
        *nid = 0;
        got_nid = 1;
    }
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int beg = v->child_start[final_id], end = v->child_start[final_id + 1];
        if (beg == end)
            break; /* a vocabulary that is only a root: not produced by the loader */
        final_id = v->child_items[beg];
        double best_d = (double)ora_descriptor_distance(feature, v->desc + (size_t)final_id * 32);
        for (int c = beg + 1; c < end; c++) {
            const int id = v->child_items[c];
            const double d = (double)ora_descriptor_distance(feature, v->desc + (size_t)id * 32);
            if (d < best_d) {
                best_d = d;
                final_id = id;
            }
        }
        if (current_level == nid_level) {
            *nid = final_id;
            got_nid = 1;
        }
    } while (v->child_start[final_id] != v->child_start[final_id + 1]);
    if (!got_nid)
        *nid = final_id;
    *word_id = v->word_id[final_id];
    *weight = v->weight[final_id];
}

static int cmp_wordfeat(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* TemplatedVocabulary.h:1140-1207.  Outputs: per-feature word / weight / node; the BowVector as ascending
 * (id, value) pairs; the FeatureVector as CSR over ascending node ids with the feature indices in insertion
 * (= feature) order.  Returns 0. */
int ora_bow_transform(const ora_vocabulary *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                      double *word_weight, int32_t *node_id, int32_t *bow_ids, double *bow_vals, int32_t *n_bow,
                      int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_items, int32_t *n_fv)
{
    for (int i = 0; i < n; i++)
        ora_vocabulary_transform_feature(v, desc + (size_t)i * 32, levelsup, &word_id[i], &word_weight[i], &node_id[i]);
    /* std::map keyed by word id / node id == stable grouping by key with the features in index order */
    int64_t *key = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++)
        if (word_weight[i] > 0) /* "not stopped", :1170 / :1198 */
            key[m++] = ((int64_t)word_id[i] << 32) | (int64_t)i;
    qsort(key, (size_t)m, sizeof(int64_t), cmp_wordfeat);
    const int tf = v->weighting == 0 /* TF_IDF */ || v->weighting == 1 /* TF */;
    int nb = 0;
    for (int p = 0; p < m;) {
        const int wid = (int)(key[p] >> 32);
        int q = p;
        double val = 0;
        for (; q < m && (int)(key[q] >> 32) == wid; q++) {
            const double w = word_weight[(int)(key[q] & 0xFFFFFFFF)];
            if (q == p)
                val = w; /* insert(value) */
            else if (tf)
                val += w; /* addWeight; addIfNotExist keeps the first */
        }
        bow_ids[nb] = wid;
        bow_vals[nb] = val;
        nb++;
        p = q;
    }
    /* ScoringObject.h:74-89: every scoring but DOT_PRODUCT normalises; L2_NORM with L2, the others with L1 */
    const int must = v->scoring != 5;
    const int l2 = v->scoring == 1;
    if (tf && nb > 0 && !must) { /* :1177-1183 */
        const double nd = (double)nb;
        for (int i = 0; i < nb; i++)
            bow_vals[i] /= nd;
    }
    if (must) { /* BowVector::normalize, BowVector.cpp:62-88 */
        double norm = 0.0;
        if (!l2) {
            for (int i = 0; i < nb; i++)
                norm += fabs(bow_vals[i]);
        } else {
            for (int i = 0; i < nb; i++)
                norm += bow_vals[i] * bow_vals[i];
            norm = sqrt(norm);
        }
        if (norm > 0.0)
            for (int i = 0; i < nb; i++)
                bow_vals[i] /= norm;
    }
    *n_bow = nb;
    /* FeatureVector */
    m = 0;
    for (int i = 0; i < n; i++)
        if (word_weight[i] > 0)
            key[m++] = ((int64_t)node_id[i] << 32) | (int64_t)i;
    qsort(key, (size_t)m, sizeof(int64_t), cmp_wordfeat);
    int nf = 0;
    for (int p = 0; p < m; p++) {
        const int nd = (int)(key[p] >> 32);
        if (p == 0 || (int)(key[p - 1] >> 32) != nd) {
            fv_nodes[nf] = nd;
            fv_start[nf] = p;
            nf++;
        }
        fv_items[p] = (int)(key[p] & 0xFFFFFFFF);
    }
    fv_start[nf] = m;
    *n_fv = nf;
    free(key);
    return 0;
}

/* rotation bin, ORBmatcher.cc:238-243 */
static int bow_rot_bin(float angle_a, float angle_b)
{
    const float factor = 1.0f / ORA_HISTO_LENGTH;
    float rot = angle_a - angle_b;
    if (rot < 0.0)
        rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORA_HISTO_LENGTH)
        bin = 0;
    return bin;
}

/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1601-1642 (on the bin counts) */
static void bow_three_maxima(const int *histo, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            *ind3 = *ind2;
            *ind2 = *ind1;
            *ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            *ind3 = *ind2;
            *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        *ind3 = -1;
    }
}

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches), ORBmatcher.cc:159-288.
 * Feature vectors as CSR (ascending node ids).  valid_kf[i]: vpMapPointsKF[i] != NULL && !isBad().
 * match_f[j] = key-frame feature whose map point frame feature j received, or -1.  Returns nmatches. */
int ora_search_by_bow(const uint8_t *desc_kf, const float *angle_kf, const uint8_t *valid_kf, int n_fv_kf,
                      const int32_t *fv_nodes_kf, const int32_t *fv_start_kf, const int32_t *fv_items_kf,
                      const uint8_t *desc_f, const float *angle_f, int nf, int n_fv_f, const int32_t *fv_nodes_f,
                      const int32_t *fv_start_f, const int32_t *fv_items_f, int th_low, float nnratio,
                      int check_orientation, int32_t *match_f)
{
    int nmatches = 0;
    int histo[ORA_HISTO_LENGTH];
    memset(histo, 0, sizeof(histo));
    int *bin_of = (int *)malloc(sizeof(int) * (size_t)(nf > 0 ? nf : 1));
    for (int j = 0; j < nf; j++) {
        match_f[j] = -1;
        bin_of[j] = -1;
    }
    int a = 0, b = 0;
    while (a < n_fv_kf && b < n_fv_f) {
        if (fv_nodes_kf[a] == fv_nodes_f[b]) {
            for (int pa = fv_start_kf[a]; pa < fv_start_kf[a + 1]; pa++) {
                const int realIdxKF = fv_items_kf[pa];
                if (valid_kf && !valid_kf[realIdxKF])
                    continue; /* :195-199 */
                const uint8_t *dKF = desc_kf + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int pb = fv_start_f[b]; pb < fv_start_f[b + 1]; pb++) {
                    const int realIdxF = fv_items_f[pb];
                    if (match_f[realIdxF] >= 0)
                        continue; /* :209-210 */
                    const int dist = ora_descriptor_distance(dKF, desc_f + (size_t)realIdxF * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= th_low) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match_f[bestIdxF] = realIdxKF;
                        if (check_orientation) {
                            const int bin = bow_rot_bin(angle_kf[realIdxKF], angle_f[bestIdxF]);
                            bin_of[bestIdxF] = bin;
                            histo[bin]++;
                        }
                        nmatches++;
                    }
                }
            }
            a++;
            b++;
        } else if (fv_nodes_kf[a] < fv_nodes_f[b]) {
            while (a < n_fv_kf && fv_nodes_kf[a] < fv_nodes_f[b]) /* lower_bound */
                a++;
        } else {
            while (b < n_fv_f && fv_nodes_f[b] < fv_nodes_kf[a])
                b++;
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        bow_three_maxima(histo, ORA_HISTO_LENGTH, &i1, &i2, &i3);
        for (int j = 0; j < nf; j++) {
            const int bn = bin_of[j];
            if (bn < 0 || bn == i1 || bn == i2 || bn == i3)
                continue;
            match_f[j] = -1;
            nmatches--;
        }
    }
    free(bin_of);
    return nmatches;
}
