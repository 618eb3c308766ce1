// On-disk vocabulary formats -> flat arrays for td_create.  Host-only code (no HIP).
//   tiktoken ".model" / ".tiktoken": one "base64(token bytes) rank" pair per line
//       (what the reference's demo reads: /root/reference/src/main.cpp:70-110, tests/throughput_test.py:137-160)
//   Hugging Face tokenizer_config.json: "added_tokens_decoder": {"<id>": {"content": "<str>", ...}, ...}
//       (reference: src/main.cpp:121-133, tests/throughput_test.py:162-180)
//   Mistral tekken.json: config.{pattern, default_vocab_size, default_num_special_tokens} and
//       vocab[i].{rank, token_bytes (base64), token_str}; id = i + default_num_special_tokens for the first
//       default_vocab_size - default_num_special_tokens entries (reference: tests/throughput_test.py:106-135)
//   the reference wrapper's own JSON files: vocabulary = [{"rank", "token_bytes": [ints], "token_string"}, ...],
//       special tokens = {"<str>": id, ...}   (reference: tokendagger/wrapper.py:116-134)
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace td {

struct TokenList {
    std::vector<uint8_t> bytes;          // concatenated token bytes
    std::vector<int64_t> offsets{0};     // [n+1]
    std::vector<int32_t> ranks;          // [n]
    int64_t size() const { return (int64_t)ranks.size(); }
    void add(const uint8_t* p, size_t len, int32_t rank) {
        bytes.insert(bytes.end(), p, p + len);
        offsets.push_back((int64_t)bytes.size());
        ranks.push_back(rank);
    }
    void clear() { bytes.clear(); offsets.assign(1, 0); ranks.clear(); }
};

struct VocabData {
    std::string pattern;
    TokenList regular, special;
    std::string err;
};

// Each loader APPENDS to v and returns false (message in v.err) on a malformed or unreadable file.
bool load_tiktoken_model(const std::string& path, VocabData& v);
bool load_hf_added_tokens(const std::string& path, VocabData& v, bool also_mergeable);
bool load_tekken_json(const std::string& path, VocabData& v);
bool load_wrapper_json(const std::string& vocab_path, const std::string& special_path, VocabData& v);

bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out);  // strict: false on a bad character / length

}  // namespace td
