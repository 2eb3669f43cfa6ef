// Host front-end for GML programs: include preprocessor, tokenizer, parser and
// stack evaluator.  It runs once per scene on the host and hands the operands of
// the `render` operator to a caller-supplied hook; nothing in here runs per ray.
//
// Behaviour (including quirks) follows the reference front-end:
//   preprocessor  src/rungml/src/main.rs:17-47
//   tokenizer     src/tokenizer.rs:75-483
//   parser        src/parser.rs:25-76
//   evaluator     src/evaluator.rs:157-848
//   transforms    src/transform.rs:43-136
//   lights (ctor) src/lights.rs:33-100
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace gml {

// ---------------------------------------------------------------- tokens
// Operator keywords, src/tokenizer.rs:2-53 (same order).
enum class Op : uint8_t {
  Acos, Addf, Addi, Apply, Asin, Clampf, Cone, Cos, Cube, Cylinder, Difference,
  Divf, Divi, Eqf, Eqi, Floor, Frac, Get, Getx, Gety, Getz, If, Intersect,
  Length, Lessf, Lessi, Light, Modi, Mulf, Muli, Negf, Negi, Plane, Point,
  Pointlight, Real, Render, Rotatex, Rotatey, Rotatez, Scale, Sin, Sphere,
  Spotlight, Sqrt, Subf, Subi, Translate, Union, Uscale,
};
const char* op_name(Op op);
bool match_operator(const std::string& s, Op* out);

struct Token {
  enum Kind : uint8_t {
    BeginFunction, EndFunction, BeginArray, EndArray,
    Identifier, Operator, Binder, Boolean, Real, Integer, Str,
  } kind;
  Op op = Op::Acos;
  bool b = false;
  double r = 0.0;
  int64_t i = 0;
  std::string s;
  bool operator==(const Token& o) const;
};

// src/tokenizer.rs:446-483.  Stops silently at the first unrecognised character.
std::vector<Token> tokenize(const std::string& text);

// ---------------------------------------------------------------- AST
struct AstNode;
using AstList = std::vector<AstNode>;
struct AstNode {
  enum Kind : uint8_t { Leaf, Array, Function } kind = Leaf;
  Token tok;                               // Leaf
  std::shared_ptr<const AstList> children; // Array / Function
};

struct GmlError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// src/parser.rs:70-76.  Throws GmlError("SyntaxError").
std::shared_ptr<const AstList> parse(const std::vector<Token>& tokens);

// ---------------------------------------------------------------- math
using Vec3 = std::array<double, 3>;
struct Mat4 { double m[4][4]; };
Mat4 mat4_identity();
// row_mat4_mul of the vecmath crate: out[i][j] = dot4(a[i], column j of b),
// products summed left to right.
Mat4 mat4_mul(const Mat4& a, const Mat4& b);

// src/transform.rs:4-7 and builders :43-136
struct Transform {
  Mat4 m = mat4_identity();
  Mat4 inv_m = mat4_identity();
  void scale(double sx, double sy, double sz);
  void translate(double tx, double ty, double tz);
  void rotatex(double deg);
  void rotatey(double deg);
  void rotatez(double deg);
};

// ---------------------------------------------------------------- scene model
enum class PrimType : uint8_t { Sphere = 0, Cube = 1, Cylinder = 2, Cone = 3, Plane = 4 };
enum class Rule : uint8_t { Union = 0, Intersect = 1, Difference = 2 };
enum class XformKind : uint8_t { Translate, Scale, Uscale, Rotatex, Rotatey, Rotatez };

struct Value;
using ValuePtr = std::shared_ptr<const Value>;

// Immutable scene DAG.  The reference mutates per-leaf transforms in place and
// forwards every transform through operators to all leaves
// (src/primitives.rs:83-102, 213-238); here a transform is a wrapper node and the
// same per-leaf sequence of builder calls is replayed when the tree is walked.
struct Node;
using NodePtr = std::shared_ptr<const Node>;
struct Node {
  enum Kind : uint8_t { Leaf, Operator, Xform } kind;
  // Leaf
  PrimType prim = PrimType::Sphere;
  ValuePtr surface;  // closure value
  // Operator
  Rule rule = Rule::Union;
  NodePtr a, b;
  // Xform
  XformKind xk = XformKind::Translate;
  double p[3] = {0, 0, 0};
  NodePtr child;
};

struct XformStep { XformKind k; double p[3]; };
void apply_step(Transform& t, const XformStep& s);

// Walks the DAG in the reference's tree order (obj1 before obj2) and calls
// leaf(prim, surface, transform) / op(rule) in postfix order.
struct TreeVisitor {
  virtual void leaf(PrimType prim, const ValuePtr& surface, const Transform& t) = 0;
  virtual void op(Rule rule) = 0;
  virtual ~TreeVisitor() {}
};
void walk_postfix(const NodePtr& root, TreeVisitor& v);

// src/lights.rs:27-100; constructor-time derived fields included.
struct Light {
  enum Kind : uint8_t { Directional = 0, Point = 1, Spot = 2 } kind;
  Vec3 dir{};     // Directional: normalize(-d).  Spot: normalize(at - pos)
  Vec3 pos{};     // Point / Spot
  Vec3 color{};
  double coscutoff = 0.0;  // Spot
  double exp = 0.0;        // Spot
};
using LightPtr = std::shared_ptr<const Light>;

// ---------------------------------------------------------------- values / env
using EnvMap = std::unordered_map<std::string, ValuePtr>;

// src/evaluator.rs:63-119 (SnapMap).  `snapshot()` copies `current` and then lets
// the entries of the old snapshot overwrite it.
struct Env {
  std::shared_ptr<const EnvMap> snap;
  EnvMap current;
  Env() : snap(std::make_shared<EnvMap>()) {}
  explicit Env(std::shared_ptr<const EnvMap> s) : snap(std::move(s)) {}
  const ValuePtr* get(const std::string& k) const;
  void insert(const std::string& k, ValuePtr v) { current[k] = std::move(v); }
  std::shared_ptr<const EnvMap> snapshot() const;
};

struct Value {
  enum Kind : uint8_t { Closure, Array, Boolean, Real, Integer, String, Point, NodeV, LightV } kind;
  // Closure
  std::shared_ptr<const EnvMap> env;
  std::shared_ptr<const AstList> ast;
  // Array (stored reversed, src/evaluator.rs:249-253)
  std::vector<ValuePtr> arr;
  bool b = false;
  double r = 0.0;
  int64_t i = 0;
  std::string s;
  Vec3 p{};
  NodePtr node;
  LightPtr light;
};
using Stack = std::vector<ValuePtr>;

ValuePtr make_int(int64_t v);
ValuePtr make_real(double v);
ValuePtr make_bool(bool v);
ValuePtr make_point(const Vec3& v);
bool value_equal(const Value& a, const Value& b);  // src/evaluator.rs:44-61

// Operands of `render`, src/evaluator.rs:749-769.
struct RenderCall {
  Vec3 amb;
  std::vector<LightPtr> lights;
  NodePtr scene;
  int64_t depth;
  double fov;
  int64_t w, h;
  std::string file;
};
using RenderHook = std::function<void(const RenderCall&)>;

// Scalar op helpers shared with the surface compiler so that constant folding
// uses the evaluator's own arithmetic (src/evaluator.rs:212-218, 329-355, 388-398, 464-472).
int64_t divi(int64_t a, int64_t b);  // throws on b == 0
int64_t modi(int64_t a, int64_t b);
double gml_acos(double x);
double gml_asin(double x);
double gml_clampf(double x);
double gml_cos(double x);
double gml_sin(double x);
int64_t gml_floor(double x);
double gml_frac(double x);

class Evaluator {
 public:
  explicit Evaluator(RenderHook hook) : hook_(std::move(hook)) {}
  // src/evaluator.rs:846-848 (`run`).  Throws GmlError.
  void run(const std::string& gml_text, Env& env, Stack& stack);
  // src/evaluator.rs:798-836
  void do_evaluate(Env& env, Stack& stack, const AstList& ast);
  // src/evaluator.rs:776-792: fresh stack [face,u,v], run, pop n ks kd color.
  void call_surface(const Value& closure, int64_t face, double u, double v,
                    Vec3* color, double* kd, double* ks, double* n);
 private:
  void eval_op(Op op, Stack& stack);
  RenderHook hook_;
};

// ---------------------------------------------------------------- preprocessor
// src/rungml/src/main.rs:17-47.  Returns the joined program text.
std::string preprocess_file(const std::string& path);

// Runs `fn` on a thread with a large stack: GML loops are recursion
// (gml/loops.ins foreach) and go tens of thousands of frames deep.
void run_with_big_stack(const std::function<void()>& fn, size_t stack_bytes = (size_t)4 << 30);

}  // namespace gml
