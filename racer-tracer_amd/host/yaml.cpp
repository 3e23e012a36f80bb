// yaml.cpp — see yaml.h.
#include "yaml.h"
#include <cctype>
#include <fstream>
#include <sstream>
#include "error.h"

namespace rthost {

bool iequals(const std::string &a, const std::string &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
    return true;
}

const YamlNode *YamlNode::find(const std::string &key) const {
    if (kind != Map) return nullptr;
    for (const auto &kv : map)
        if (iequals(kv.first, key)) return &kv.second;
    return nullptr;
}

namespace {

struct Line {
    int indent;
    std::string text; // without indentation, comments and trailing blanks
    int number;
};

struct Parser {
    std::string file;
    std::vector<Line> lines;

    [[noreturn]] void fail(int line, const std::string &why) const {
        throw TracerError::Configuration(file, "line " + std::to_string(line) + ": " + why);
    }

    static std::string trim(const std::string &s) {
        size_t b = 0, e = s.size();
        while (b < e && std::isspace((unsigned char)s[b])) ++b;
        while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
        return s.substr(b, e - b);
    }

    // Drop a trailing comment: '#' at the start or after whitespace, outside quotes.
    static std::string strip_comment(const std::string &s) {
        char quote = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            char c = s[i];
            if (quote) {
                if (c == quote) quote = 0;
            } else if (c == '"' || c == '\'') {
                quote = c;
            } else if (c == '#' && (i == 0 || std::isspace((unsigned char)s[i - 1]))) {
                return s.substr(0, i);
            }
        }
        return s;
    }

    void split(const std::string &text) {
        std::istringstream in(text);
        std::string raw;
        int n = 0;
        while (std::getline(in, raw)) {
            ++n;
            if (!raw.empty() && raw.back() == '\r') raw.pop_back();
            std::string body = strip_comment(raw);
            size_t ind = 0;
            while (ind < body.size() && body[ind] == ' ') ++ind;
            if (ind < body.size() && body[ind] == '\t') fail(n, "tab used for indentation");
            std::string t = trim(body);
            if (t.empty() || t == "---" || t == "...") continue;
            lines.push_back(Line{(int)ind, t, n});
        }
    }

    // Position of the ':' that ends a block-mapping key, or npos.
    static size_t key_colon(const std::string &t) {
        char quote = 0;
        int depth = 0;
        for (size_t i = 0; i < t.size(); ++i) {
            char c = t[i];
            if (quote) {
                if (c == quote) quote = 0;
                continue;
            }
            if (c == '"' || c == '\'') quote = c;
            else if (c == '[' || c == '{') ++depth;
            else if (c == ']' || c == '}') --depth;
            else if (c == ':' && depth == 0 && (i + 1 == t.size() || t[i + 1] == ' ')) return i;
        }
        return std::string::npos;
    }

    static std::string unquote(const std::string &s) {
        if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
            return s.substr(1, s.size() - 2);
        return s;
    }

    // ---- flow style: [a, b], {k: v} ----
    void skip_ws(const std::string &s, size_t &i) const {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    }

    YamlNode flow_value(const std::string &s, size_t &i, int line, bool in_map_key) const {
        skip_ws(s, i);
        YamlNode n;
        n.line = line;
        if (i >= s.size()) return n;
        if (s[i] == '[') {
            n.kind = YamlNode::Seq;
            ++i;
            for (;;) {
                skip_ws(s, i);
                if (i >= s.size()) fail(line, "unterminated '['");
                if (s[i] == ']') { ++i; break; }
                n.seq.push_back(flow_value(s, i, line, false));
                skip_ws(s, i);
                if (i < s.size() && s[i] == ',') ++i;
                else if (i < s.size() && s[i] == ']') { ++i; break; }
                else fail(line, "expected ',' or ']'");
            }
            return n;
        }
        if (s[i] == '{') {
            n.kind = YamlNode::Map;
            ++i;
            for (;;) {
                skip_ws(s, i);
                if (i >= s.size()) fail(line, "unterminated '{'");
                if (s[i] == '}') { ++i; break; }
                YamlNode k = flow_value(s, i, line, true);
                skip_ws(s, i);
                if (i >= s.size() || s[i] != ':') fail(line, "expected ':' in flow mapping");
                ++i;
                n.map.emplace_back(k.scalar, flow_value(s, i, line, false));
                skip_ws(s, i);
                if (i < s.size() && s[i] == ',') ++i;
                else if (i < s.size() && s[i] == '}') { ++i; break; }
                else fail(line, "expected ',' or '}'");
            }
            return n;
        }
        n.kind = YamlNode::Scalar;
        if (s[i] == '"' || s[i] == '\'') {
            char q = s[i];
            size_t e = s.find(q, i + 1);
            if (e == std::string::npos) fail(line, "unterminated quoted string");
            n.scalar = s.substr(i + 1, e - i - 1);
            i = e + 1;
            return n;
        }
        size_t b = i;
        while (i < s.size() && s[i] != ',' && s[i] != ']' && s[i] != '}' && !(in_map_key && s[i] == ':')) ++i;
        n.scalar = trim(s.substr(b, i - b));
        return n;
    }

    YamlNode inline_value(const std::string &text, int line) const {
        std::string t = trim(text);
        YamlNode n;
        n.line = line;
        if (t.empty() || t == "~" || t == "null") return n;
        if (t[0] == '[' || t[0] == '{') {
            size_t i = 0;
            n = flow_value(t, i, line, false);
            skip_ws(t, i);
            if (i != t.size()) fail(line, "trailing characters after flow collection");
            return n;
        }
        n.kind = YamlNode::Scalar;
        n.scalar = unquote(t);
        return n;
    }

    // ---- block style ----
    YamlNode block(size_t &idx, int indent) {
        const Line &first = lines[idx];
        if (first.text.rfind("- ", 0) == 0 || first.text == "-") return sequence(idx, indent);
        if (key_colon(first.text) != std::string::npos) return mapping(idx, indent);
        YamlNode n = inline_value(first.text, first.number); // plain scalar on its own line
        ++idx;
        return n;
    }

    YamlNode mapping(size_t &idx, int indent) {
        YamlNode n;
        n.kind = YamlNode::Map;
        n.line = lines[idx].number;
        while (idx < lines.size() && lines[idx].indent == indent) {
            const Line &ln = lines[idx];
            if (ln.text.rfind("- ", 0) == 0) break;
            size_t colon = key_colon(ln.text);
            if (colon == std::string::npos) fail(ln.number, "expected 'key: value'");
            std::string key = unquote(trim(ln.text.substr(0, colon)));
            std::string rest = trim(ln.text.substr(colon + 1));
            ++idx;
            YamlNode value;
            value.line = ln.number;
            if (!rest.empty()) {
                value = inline_value(rest, ln.number);
            } else if (idx < lines.size() && lines[idx].indent > indent) {
                value = block(idx, lines[idx].indent);
            }
            for (const auto &kv : n.map)
                if (kv.first == key) fail(ln.number, "duplicate key '" + key + "'");
            n.map.emplace_back(key, std::move(value));
        }
        if (idx < lines.size() && lines[idx].indent > indent) fail(lines[idx].number, "unexpected indentation");
        return n;
    }

    YamlNode sequence(size_t &idx, int indent) {
        YamlNode n;
        n.kind = YamlNode::Seq;
        n.line = lines[idx].number;
        while (idx < lines.size() && lines[idx].indent == indent &&
               (lines[idx].text.rfind("- ", 0) == 0 || lines[idx].text == "-")) {
            Line ln = lines[idx];
            std::string rest = ln.text.size() > 1 ? trim(ln.text.substr(2)) : std::string();
            if (rest.empty()) {
                ++idx;
                if (idx < lines.size() && lines[idx].indent > indent) n.seq.push_back(block(idx, lines[idx].indent));
                else n.seq.push_back(YamlNode());
            } else if (rest[0] != '[' && rest[0] != '{' && key_colon(rest) != std::string::npos) {
                // "- key: value": a mapping whose first entry shares the dash line
                int item_indent = indent + 2;
                lines[idx].indent = item_indent;
                lines[idx].text = rest;
                n.seq.push_back(mapping(idx, item_indent));
            } else {
                n.seq.push_back(inline_value(rest, ln.number));
                ++idx;
            }
        }
        return n;
    }
};

} // namespace

YamlNode parse_yaml(const std::string &text, const std::string &file_name) {
    Parser p;
    p.file = file_name;
    p.split(text);
    if (p.lines.empty()) return YamlNode();
    size_t idx = 0;
    YamlNode root = p.block(idx, p.lines[0].indent);
    if (idx != p.lines.size()) p.fail(p.lines[idx].number, "unexpected content (bad indentation?)");
    return root;
}

YamlNode parse_yaml_file(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw TracerError::Configuration(path, "configuration file \"" + path + "\" not found");
    std::ostringstream ss;
    ss << in.rdbuf();
    return parse_yaml(ss.str(), path);
}

} // namespace rthost
